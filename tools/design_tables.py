"""Prints the markdown tables of DESIGN.md sections 4.2 / 4.3 / 5 from the committed measurements under profiles/ (round tag as
argument, default r06): per-class table of the headline step (HIP-event records + PMC summary + in-run traffic), dominant class
and whole step of the other workloads, results table against the previous round.   python tools/design_tables.py [r06] [r05]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
PREV = sys.argv[2] if len(sys.argv) > 2 else "r05"
P = lambda n: os.path.join(ROOT, "profiles", n)
load = lambda n: json.load(open(P(n))) if os.path.exists(P(n)) else None
last_line = lambda n: json.loads([l for l in open(P(n)) if l.startswith("{")][-1]) if os.path.exists(P(n)) else None

KIND = [("conv_", "3×3 convolutions `conv_halo_x3`"), ("tail_", "56×56 tails `chain_x3` (3×3 + expand + identity + next reduce)"),
        ("chainw_", "28×28 / 14×14 chains `chain_x3w` (expand + identity + next reduce)"), ("dual_", "expand + downsample branch `gemm_x3` dual form"),
        ("gemm_f16x3", "remaining 1×1 products `gemm_x3`"), ("conv1_", "stem + max-pool + first reduce `conv1_pool_x3`")]


def headline():
    recs = load(f"{R}_pipeline_f16x3_bench_kernels.json")
    pmc = (load(f"{R}_pipeline_f16x3_pmc.json") or {}).get("classes", {})
    fin = last_line(f"{R}_bench_final.json") or {}
    by = (fin.get("roofline") or {}).get("traffic_by_class", {})
    slides = recs[0]["count"] / 6 if recs and recs[0]["name"].startswith("conv_f16x3_M196000") else 24      # 6 such launches per slide
    x3 = [r for r in recs if "f16x3" in r["name"]]
    tot = sum(r["total_ms"] for r in x3) / slides
    print("| Kernel class (1000 patches per launch) | launches per slide | µs each | ms per slide (share) | algorithmic rate | of the 833 TF ceiling | L2-miss bytes vs algorithmic | MFMA busy |")
    print("|---|---|---|---|---|---|---|---|")
    for r in x3:
        us = r["total_ms"] / r["count"] * 1e3
        ms = r["total_ms"] / slides
        tf = r["flops"] / us / 1e6
        gbs = r["bytes"] / us / 1e3
        c = pmc.get(r["name"]) or pmc.get(r["name"].replace("_P3136000", "")) or {}
        busy = f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] * 8 / (c['GRBM_GUI_ACTIVE'] * 1024):.2f}" if c.get("GRBM_GUI_ACTIVE") else "—"
        t = by.get(r["name"])
        tr = f"{t['hbm_bytes'] / 1e9:.2f} / {t['algorithmic_bytes'] / 1e9:.2f} GB = {t['ratio']:.2f}×" if t and t.get("ratio") else \
            (f"{c['hbm_bytes_avg'] / 1e9:.2f} / {r['bytes'] / 1e9:.2f} GB = {c['hbm_bytes_avg'] / r['bytes']:.2f}×" if c.get("hbm_bytes_avg") else "—")
        print(f"| `{r['name']}` | {r['count'] / slides:.0f} | {us:.0f} | {ms:.2f} ({ms / tot:.1%}) | {tf:.0f} TF, {gbs / 1e3:.2f} TB/s | {tf / 833.3:.2f} | {tr} | {busy} |")
    print(f"\nPer slide {tot:.1f} ms of split-mode kernel time: " + ", ".join(
        f"{label} {sum(r['total_ms'] for r in x3 if r['name'].startswith(pre)) / slides:.2f}" for pre, label in KIND) + ".")


def others():
    print("| Workload | step | dominant class (share of the step) | its rate | of peak | L2-miss bytes vs algorithmic | whole step of peak |")
    print("|---|---|---|---|---|---|---|")
    for tag, name in (("vis_train_bf16", "ViS training step, B = 64, bf16 (config 2)"), ("spatial_bf16", "50 000-tile sliding-window slide, bf16 (config 5)"),
                      ("pipeline_uni_bf16", "pipeline with the UNI ViT-L/16 embedder, bf16")):
        d = last_line(f"{R}_{tag}_bench_line.json")
        if not d:
            continue
        r = d["roofline"]
        tr = f"{r['traffic'] / 1e6:.0f} / {r['algorithmic_bytes'] / 1e6:.0f} MB = {r['traffic'] / r['algorithmic_bytes']:.2f}×" + ("" if r.get("traffic_measured") else " (committed pass)") if r.get("traffic") else "—"
        print(f"| {name} | {d['ms_per_step']:.2f} ms ({d['value']:.3g} slides/s) | `{r['kernel']}` × {r['launches_per_step']} ({r['share_of_instrumented_time']:.0%}) | "
              f"{r['achieved']:.0f} {r['unit']} in {r['avg_us']:.1f} µs | {r['frac']:.3f} | {tr} | {r.get('frac_end_to_end', '—')} |")


def results():
    a, b = last_line(f"{R}_bench_final.json"), last_line(f"{PREV}_bench_final.json")
    if not a:
        return
    sa, sb = a.get("secondary", {}), (b or {}).get("secondary", {})
    row = lambda name, va, vb: print(f"| {name} | {vb} | {va} |")
    print(f"| Workload (one MI355X, `profiles/{R}_bench_final.json` vs `{PREV}_bench_final.json`; boxes of the pool differ by 1–4 %) | round {PREV[1:].lstrip('0')} | round {R[1:].lstrip('0')} |")
    print("|---|---|---|")
    f = lambda d, k="value", fmt="{:.2f}": fmt.format(d[k]) if d and k in d else "—"
    row("**pipeline `f16x3`, from pinned host (headline)**, slides/s", f"**{a['value']:.2f}** ({a['power']['energy_j_per_slide']:.1f} J per slide at {a['power']['package_w']:.0f} W, sclk {a['power']['sclk_frac_of_max']:.2f} of max; "
        f"end to end {a['roofline']['end_to_end']['frac_of_mfma_peak']:.3f} of the mode's ceiling)", f(b))
    for key, name in (("pipeline_resident_in_hbm", "pipeline `f16x3`, patches resident in HBM"), ("pipeline_fp32_exact_mfma_mode", "pipeline, exact-fp32 MFMA mode (`reference_arithmetic`)"),
                      ("pipeline_256px_patches", "pipeline `f16x3`, 256-px patches"), ("pipeline_bf16_throughput_mode_from_pinned_host", "pipeline bf16 throughput mode (not parity-grade)"),
                      ("pipeline_uni_vit_l16_embedder", "pipeline with the UNI ViT-L/16 embedder")):
        row(name + ", slides/s", f(sa.get(key)), f(sb.get(key)))
    row("`vis_train` bf16 (config 2), ms per step", f(sa.get("vis_train_bf16"), "ms_per_step", "{:.3f}"), f(sb.get("vis_train_bf16"), "ms_per_step", "{:.3f}"))
    row("`train_kfold`, 64 slides per GPU (config 4's one-GPU share), ms per step", f(sa.get("train_kfold_64_slides_per_gpu"), "ms_per_step", "{:.1f}"), f(sb.get("train_kfold_64_slides_per_gpu"), "ms_per_step", "{:.1f}"))
    row("`spatial` 50 000 tiles (config 5), s per slide", f(sa.get("spatial_50k_tiles"), "ms_per_step", "{:.1f} ms"), f(sb.get("spatial_50k_tiles"), "ms_per_step", "{:.1f} ms"))
    cpu = lambda d: (f"{d['cpu_baseline']['value']:.4f} literal / {d['cpu_baseline'].get('batched_value', float('nan')):.4f} batched ({d['cpu_baseline']['cores']} cores)"
                     if d and d.get("cpu_baseline") else "—")
    row("CPU baseline (config 1 restated on the host cores, same run), slides/s", cpu(a), cpu(b))
    acc = a.get("accuracy_vs_reference", {})
    print("\nAccuracy of the headline mode on the four reference-made slides (`accuracy_vs_reference`): " + "; ".join(
        f"{k}: features {v['feature_rel_err']:.1e}, labels {v['labels_equal']}/1000, prediction {v['prediction_rel_err']:.1e}" for k, v in acc.items() if isinstance(v, dict) and "labels_equal" in v) + ".")


if __name__ == "__main__":
    print("<!-- 4.2 -->"); headline()
    print("\n<!-- 4.3 -->"); others()
    print("\n<!-- 5 -->"); results()
