#!/usr/bin/env python
"""bench.py -- slides/sec of the SEQUOIA hot path on MI355X (driver contract in the task brief).

    python bench.py --gpus N --steps K --warmup W [--workload pipeline|vis_train|vis_fwd|train_kfold|spatial] [--dtype bf16|fp32]

Default workload = the headline configuration of BASELINE.json's metric (config 3) AS STATED: 1000 x 224 x 224 uint8
patches per slide, streamed from pinned host memory every step -> ResNet-50 embed -> k-Means(100) -> ViS forward, in the
parity-grade mode (``f16x3``: fp16 hi/lo planes, three MFMAs per product, fp32 accumulation -- the reference's fp32
results to ~1e-6, cluster labels bit-equal).  The line carries ``accuracy_vs_reference`` (the golden slide of
tests/golden/pipeline_slide.npz through the same pipeline in the same mode) and ``secondary``: the resident twin (upload
outside the timed region), the plain-bf16 throughput mode with ITS accuracy figures, the exact-fp32 mode, BASELINE
config 2 (``vis_train``), the UNI embedder and config 5, each timed the same way in a fresh process.

One process per GPU: with --gpus N > 1 and no WORLD_SIZE in the environment bench.py starts the N ranks itself
(torch.distributed.run); slides are independent units, so ranks shard them with no data-path collective.  Only the
training workloads have an exchange step: the RCCL all-reduce of the flat gradient buffer.

Prints ONE JSON line on rank 0 with the headline value plus:
  roofline     -- dominant kernel, algorithmic FLOP (or bytes) per launch / HIP-event duration,
                  measured on the launch stream in extra instrumented steps after the timed region
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's torch-CPU path) timed on
                  this host's cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sequoia_pub_amd  # noqa: E402,F401
from sequoia_pub_amd import _lib, synth  # noqa: E402

PEAK = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3, "f16x3": 2500.0 / 3}     # dense MFMA TFLOP/s (MI355X_MICROARCH.md); bf16x3: three bf16 MFMAs per product
HBM_PEAK_GBS = 8000.0

VIS_CFG = dict(num_outputs=20820, input_dim=1024, depth=6, nheads=16,
               dimensions_f=64, dimensions_s=64, dimensions_c=64)


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


class PowerSampler:
    """Package power and shader clock of this rank's GPU while the timed region runs (rank 0 only): `rocm-smi` from a second thread
    every ~0.7 s.  The split-fp16 pipeline runs AT the 1400 W cap with the clock pulled to ~1.9 GHz: a rate quoted
    against the 2.4 GHz peak says less than the same rate against the clock the chip actually sustains."""
    MAX_SCLK_MHZ = 2400.0

    def __init__(self, gpu_index):
        import shutil
        import threading
        self.gpu = gpu_index
        self.samples = []
        self.stop = threading.Event()
        self.th = threading.Thread(target=self._run, daemon=True) if shutil.which("rocm-smi") else None

    def _read(self):
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.gpu), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        except Exception:
            return None
        p = re.search(r"Package Power \(W\): ([0-9.]+)", out)
        c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        return (float(p.group(1)), float(c.group(1))) if p and c else None

    def _run(self):
        while not self.stop.is_set():
            v = self._read()
            if v and not self.stop.is_set():
                self.samples.append(v)
            self.stop.wait(0.7)

    def __enter__(self):
        if self.th:
            self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.th:
            self.th.join(timeout=15)

    def report(self):
        s = self.samples[1:] if len(self.samples) > 2 else self.samples      # rocm-smi's power is a running average: drop the first reading
        if not s:
            return None
        return {"package_w": round(sum(v[0] for v in s) / len(s), 1), "sclk_mhz": round(sum(v[1] for v in s) / len(s), 1),
                "sclk_frac_of_max": round(sum(v[1] for v in s) / len(s) / self.MAX_SCLK_MHZ, 3), "samples": len(s),
                "source": "rocm-smi --showpower --showclocks during the timed region (package power cap 1400 W)"}


LAST_POWER = [None]


def timed_region(step_fn, steps, warmup, world, device, flush_fn=None):
    for _ in range(warmup):
        step_fn()
    if flush_fn:
        flush_fn()
    barrier_sync(world)
    rank = int(os.environ.get("RANK", "0"))
    sampler = PowerSampler(device.index if hasattr(device, "index") and device.index is not None else 0) if rank == 0 and not os.environ.get("SQ_BENCH_NO_POWER") else None
    if sampler:
        sampler.__enter__()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
        if os.environ.get("SQ_BENCH_STEPTIMES"):       # debugging aid: per-step wall time (adds a sync per step)
            torch.cuda.synchronize()
            print(f"step done at {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr)
    if flush_fn:                      # streaming workloads: work still in flight belongs to the timed region
        flush_fn()
    barrier_sync(world)
    dt = time.perf_counter() - t0
    if sampler:
        sampler.__exit__()
        LAST_POWER[0] = sampler.report()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def roofline_from_profile(step_fn, steps, dtype_name, workload_name=""):
    """Run `steps` more steps with HIP-event instrumentation on and pick the dominant kernel.

    The timed region overlaps independent kernels on helper streams (weight gradients beside the dX chain, two ResNet
    chains, ...), which stretches every kernel's own start-to-end time without saying anything about the kernel.
    For the roofline the helper streams are therefore switched off during these extra steps (same kernels, same
    shapes, one at a time); the rocprofv3 summary taken the same way is profiles/r03_*_kernel_stats_serial.csv, the one
    of the overlapped timed region profiles/r03_*_kernel_stats.csv."""
    serial = {"SQ_BWD_ONE_STREAM": "1", "SQ_FWD_ONE_STREAM": "1", "SQ_RESNET_STREAMS": "1", "SQ_SPATIAL_STREAMS": "1"}
    saved = {k: os.environ.get(k) for k in serial}
    os.environ.update(serial)
    try:
        step_fn()                      # untimed: lets workspaces of the serial path settle
        torch.cuda.synchronize()
        marker_file = os.environ.get("SQ_PROF_MARKERS")       # set by measure_traffic_marked in its rocprofv3 child passes
        _lib.prof_enable(True, markers=bool(marker_file))
        for _ in range(steps):
            step_fn()
        recs = _lib.prof_report()
        _lib.prof_enable(False)
        if marker_file:
            with open(marker_file, "w") as f:
                json.dump(_lib.prof_marker_names(), f)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if not recs:
        return None, []
    dom = max(recs, key=lambda r: r["total_ms"])
    avg_s = dom["total_ms"] / dom["count"] * 1e-3
    total_ms = sum(r["total_ms"] for r in recs)
    flops_per_byte = dom["flops"] / max(dom["bytes"], 1.0)
    ridge = PEAK[dtype_name] * 1e12 / (HBM_PEAK_GBS * 1e9)
    if flops_per_byte >= ridge:
        ach = dom["flops"] / avg_s / 1e12
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK[dtype_name], "unit": "TFLOP/s",
                "frac": round(ach / PEAK[dtype_name], 4)}
    else:
        ach = dom["bytes"] / avg_s / 1e9
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4)}
    # both roofs, whichever side of the ridge the algorithmic intensity puts the label on (operands that live in L2 / the memory-side
    # cache make a product just below the ridge look HBM-bound when it is not)
    roof["both_roofs"] = {"tflops": round(dom["flops"] / avg_s / 1e12, 2), "mfma_frac": round(dom["flops"] / avg_s / 1e12 / PEAK[dtype_name], 4),
                          "gbs": round(dom["bytes"] / avg_s / 1e9, 1), "hbm_frac": round(dom["bytes"] / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                          "flops_per_byte": round(flops_per_byte, 1), "ridge": round(ridge, 1)}
    roof.update({"traffic": None, "kernel": dom["name"], "avg_us": round(avg_s * 1e6, 2),
                 "launches_per_step": dom["count"] // max(steps, 1),
                 "share_of_instrumented_time": round(dom["total_ms"] / total_ms, 3),
                 "algorithmic_bytes": round(dom["bytes"]), "algorithmic_flops": round(dom["flops"]),
                 "timed": "helper streams off during the profiled steps (kernels one at a time)"})
    # HBM-side bytes per launch of this kernel/shape from the committed rocprofv3 PMC passes of the same command
    # (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes; tools/pmc_summary.py)
    roof["traffic_measured"] = False          # read from the committed PMC passes of the same command, not from this run
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        pmc = os.path.join(ROOT, "profiles", f"{rnd}_{workload_name}_{dtype_name}_pmc.json")
        if os.path.exists(pmc):
            cls = json.load(open(pmc)).get("classes", {}).get(dom["name"])
            if cls and "hbm_bytes_avg" in cls:
                roof["traffic"] = cls["hbm_bytes_avg"]
                roof["traffic_source"] = os.path.relpath(pmc, ROOT)
                break
    return roof, recs


def _kernel_of_class(name):
    """(kernel symbol substring, grid size in threads) of a profiled kernel class of the split-mode ResNet, from its name; None if
    the class is not one whose launch geometry follows from the name."""
    import re
    m = re.match(r"(conv|gemm)_(f16x3|bf16x3)_M(\d+)_N(\d+)_K(\d+)$", name)
    if m:
        kind, f16, M, N, K = m.group(1), m.group(2) == "f16x3", int(m.group(3)), int(m.group(4)), int(m.group(5))
        if kind == "conv":
            return "conv_halo_x3_kernel", -(-M // 256) * (N // 128 if N % 128 == 0 else -(-N // 64)) * 512
        if K >= 512 and N > 64:
            return "gemm_x3_kernel<256", -(-M // 256) * -(-N // 128) * 512
        return "gemm_x3_kernel<128", -(-M // 128) * -(-N // (128 if N > 64 else 64)) * 256
    m = re.match(r"dual_(f16x3|bf16x3)_M(\d+)_N(\d+)_K(\d+)_K(\d+)$", name)
    if m:
        f16 = "true" if m.group(1) == "f16x3" else "false"
        return f"gemm_x3_kernel<128, 2, false, {f16}, false, true>", -(-int(m.group(2)) // 128) * (int(m.group(3)) // 128) * 256
    m = re.match(r"(tail|chain)_(f16x3|bf16x3)_c64_cn(\d+)(_ds)?_P(\d+)$", name)
    if m:
        f16 = "true" if m.group(2) == "f16x3" else "false"
        return (f"chain_x3_kernel<{m.group(3)}, {f16}, {'true' if m.group(4) else 'false'}, {'true' if m.group(1) == 'tail' else 'false'}",
                -(-int(m.group(5)) // 64) * 256)
    m = re.match(r"chainw_(f16x3|bf16x3)_c(\d+)_cn(\d+)_P(\d+)$", name)
    if m:
        return f"chain_x3w_kernel<{m.group(2)}, {m.group(3)}, {'true' if m.group(1) == 'f16x3' else 'false'}", -(-int(m.group(4)) // 128) * 512
    if name in ("conv1_pool_f16x3", "conv1_pool_bf16x3", "conv1_pool_reduce_f16x3", "conv1_pool_reduce_bf16x3"):
        return "conv1_pool_x3_kernel", 256 * 512
    return None


def measure_traffic(roof, args, recs=()):
    """HBM-side bytes per launch from the PMC counters of THIS build: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE cannot share a
    pass: TCC slots) over one one-slide step of the same workload in child processes, corrected as MI355X_MICROARCH.md prescribes
    (FETCH_SIZE counts 64 B per 128-byte request on gfx950: doubled; KiB units).  `roofline.traffic` = the dominant class;
    `roofline.traffic_by_class` = every split-mode class whose launch geometry follows from its name (plain and dual products,
    halo-staged 3x3, the 56x56 tails, the 28x28 / 14x14 chains, the stem) with its algorithmic bytes and the wasted-traffic ratio.
    Runs by default for the default workload when rocprofv3 is on the box (--no-measure-traffic: the committed pass instead)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return False
    want = {}
    for n in [r["name"] for r in recs] + [roof.get("kernel", "")]:
        k = _kernel_of_class(n)
        if k:
            want[n] = k
    if roof.get("kernel") not in want:
        return False
    alg = {r["name"]: r["bytes"] for r in recs}        # prof records carry bytes PER LAUNCH
    alg.setdefault(roof["kernel"], roof.get("algorithmic_bytes", 0))
    got = {c: {} for c in ("FETCH_SIZE", "WRITE_SIZE")}
    tmp = tempfile.mkdtemp(prefix="sq_pmc_", dir="/tmp")
    t0 = time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__), "--workload", args.workload, "--dtype", args.dtype, "--slides", "1", "--steps", "1", "--warmup", "0",
                   "--patches", str(args.patches), "--patch-size", str(args.patch_size), "--sub-batch", str(args.sub_batch),
                   "--no-secondary", "--no-cpu-baseline", "--no-accuracy", "--no-measure-traffic"]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
            rows = []
            for f in glob.glob(out + "/**/*_counter_collection.csv", recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
            for name, (symbol, grid) in want.items():
                vals = [float(r["Counter_Value"]) for r in rows if symbol in r["Kernel_Name"] and int(r["Grid_Size"]) == grid]
                if vals:
                    got[counter][name] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    by = {}
    for name in want:
        if name in got["FETCH_SIZE"] and name in got["WRITE_SIZE"]:
            meas = got["FETCH_SIZE"][name] * 1024 * 2 + got["WRITE_SIZE"][name] * 1024
            twins = sorted(n for n in want if n != name and want[n] == want[name])      # same symbol and grid: the counters cannot tell them apart
            by[name] = {"hbm_bytes": round(meas), "algorithmic_bytes": round(alg.get(name, 0)),
                        "ratio": round(meas / alg[name], 3) if alg.get(name) and not twins else None}
            if twins:
                by[name]["mean_over_launches_shared_with"] = twins
    if roof["kernel"] not in by:
        return False
    roof["traffic"] = by[roof["kernel"]]["hbm_bytes"]
    roof["traffic_measured"] = True
    roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate one-slide passes of this build in child processes, "
                              f"{round(time.time() - t0, 1)} s")
    roof["traffic_by_class"] = by
    # what the counters see: requests that leave an XCD's L2.  Re-reads another XCD fetched a moment ago are served by the 256 MB
    # memory-side cache, not by DRAM -- a ratio above 1 is fabric traffic (operand panels / halo rows re-read under a second L2),
    # not necessarily DRAM traffic (the 56x56 tails went from 1.17-1.27 to 1.01-1.03 at an unchanged rate)
    roof["traffic_note"] = "L2-miss bytes at the XCDs (FETCH_SIZE / WRITE_SIZE): includes re-reads the memory-side cache serves"
    return True


def measure_traffic_marked(roof, args, recs=()):
    """`roofline.traffic` for ANY workload, without knowing kernel symbols or launch geometry: two rocprofv3 passes (FETCH_SIZE,
    WRITE_SIZE: separate passes, gfx950 corrections as in measure_traffic) over one step of the same workload in child processes
    whose instrumented launches are bracketed by marker launches (sq_prof_enable(2): the marker's grid size is the class number).
    In the dispatch-ordered counter table every dispatch between an opening marker and the closing one belongs to that class."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return False
    alg = {r["name"]: r["bytes"] for r in recs}
    tmp = tempfile.mkdtemp(prefix="sq_pmcm_", dir="/tmp")
    t0 = time.time()
    spans = {c: {} for c in ("FETCH_SIZE", "WRITE_SIZE")}          # counter -> class -> [per-launch sums]
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out, names_file = os.path.join(tmp, counter), os.path.join(tmp, counter + "_names.json")
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__), "--workload", args.workload, "--dtype", args.dtype, "--steps", "1", "--warmup", "0",
                   "--batch", str(args.batch), "--epochs", str(args.epochs), "--slides", str(1 if args.workload == "pipeline" else args.slides),
                   "--patches", str(args.patches), "--patch-size", str(args.patch_size), "--sub-batch", str(args.sub_batch),
                   "--uni-sub-batch", str(args.uni_sub_batch), "--grid", str(args.grid[0]), str(args.grid[1]), "--batch-windows", str(args.batch_windows),
                   "--embedder", args.embedder, "--no-secondary", "--no-cpu-baseline", "--no-accuracy", "--no-measure-traffic"]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", SQ_PROF_MARKERS=names_file, SQ_BENCH_NO_POWER="1"),
                           capture_output=True, text=True, timeout=900)
            if not os.path.exists(names_file):
                return False
            names = json.load(open(names_file))
            rows = []
            for f in glob.glob(out + "/**/*_counter_collection.csv", recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            cur, acc = None, 0.0
            for r in rows:
                if "sq_prof_marker_kernel" in r["Kernel_Name"]:
                    blocks = int(r["Grid_Size"]) // 64
                    if cur is not None:
                        spans[counter].setdefault(cur, []).append(acc)
                    cur, acc = (names[blocks - 2] if 2 <= blocks < len(names) + 2 else None), 0.0
                elif cur is not None:
                    acc += float(r["Counter_Value"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    by = {}
    for name in spans["FETCH_SIZE"]:
        f, w = spans["FETCH_SIZE"][name], spans["WRITE_SIZE"].get(name)
        if not f or not w:
            continue
        meas = sum(f) / len(f) * 1024 * 2 + sum(w) / len(w) * 1024          # FETCH_SIZE counts 64 B per 128-byte request on gfx950; KiB units
        by[name] = {"hbm_bytes": round(meas), "algorithmic_bytes": round(alg.get(name, 0)), "launches_seen": len(f),
                    "ratio": round(meas / alg[name], 3) if alg.get(name) else None}
    if roof.get("kernel") not in by:
        return False
    roof["traffic"] = by[roof["kernel"]]["hbm_bytes"]
    roof["traffic_measured"] = True
    roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes over one step of this workload in child processes, "
                              f"dispatches attributed to classes by marker launches (sq_prof_enable(2)), {round(time.time() - t0, 1)} s")
    top = sorted(by, key=lambda n: -next((r["total_ms"] for r in recs if r["name"] == n), 0.0))[:12]
    roof["traffic_by_class"] = {n: by[n] for n in top}
    roof["traffic_note"] = "L2-miss bytes at the XCDs (FETCH_SIZE / WRITE_SIZE): includes re-reads the memory-side cache serves"
    return True


def timed_cpu_sample(fn, units_per_call, budget_s=12.0, candidates=(8, 16, 32, 64, 128)):
    """Time `fn` (a CPU oracle call) on this host: try a few torch thread counts once each (the
    reference's many small per-head GEMMs do not scale to every core of a big host), keep the
    fastest, then repeat it until ~budget_s seconds of CPU work.  Returns (units/s, threads, reps)."""
    ncpu = os.cpu_count() or 1
    best = None
    for t in [c for c in candidates if c <= ncpu] or [ncpu]:
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, t)
        if dt > budget_s:          # already slow: do not try more settings
            break
    torch.set_num_threads(best[1])
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or time.perf_counter() - t0 < budget_s:
        fn()
        reps += 1
    dt = time.perf_counter() - t0
    return units_per_call * reps / dt, best[1], reps


# ------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------
def make_vis(dtype_name, device):
    from sequoia_pub_amd.vis import ViS
    torch.manual_seed(99)
    model = ViS(**VIS_CFG, num_clusters=100, device=str(device), compute_dtype=dtype_name)
    model.to(device)
    return model


def workload_vis_fwd(args, rank, world, device):
    """ViS forward on pre-computed cluster tokens (BASELINE config 2 model/shape, inference)."""
    B = args.batch
    model = make_vis(args.dtype, device).eval()
    x = torch.from_numpy(synth.cluster_tokens(99 + rank, B, 1024)).to(device)

    def step():
        with torch.no_grad():
            model(x)

    def cpu_baseline():
        from oracle import vis_oracle
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        nb = 8
        xc = x[:nb].cpu()

        def call():
            with torch.no_grad():
                vis_oracle.vis_forward(sd, xc)
        rate, threads, reps = timed_cpu_sample(call, nb)
        return {"value": round(rate, 3), "unit": "slides/s", "cores": threads, "kind": "port",
                "sample": f"oracle.vis_oracle.vis_forward (torch-CPU fp32), {reps} x batch {nb} of the same tokens"}

    return dict(step=step, slides_per_step=B, cpu_baseline=cpu_baseline,
                config={"workload": "vis_fwd: ViS(D=1024, depth 6, 16 heads, G=20820) forward on 100 cluster tokens/slide",
                        "batch_per_gpu": B, "parallelism": f"slide-sharded x{world}"})


def workload_vis_train(args, rank, world, device):
    """BASELINE config 2: lin-attn (ViS) forward + backward + AdamW, 20k genes, batch 64 slides per GPU."""
    from sequoia_pub_amd import train as sq_train
    B = args.batch
    model = make_vis(args.dtype, device).train()
    x = torch.from_numpy(synth.cluster_tokens(99 + rank, B, 1024)).to(device)
    y = torch.from_numpy(synth.rna_targets(199 + rank, B, VIS_CFG["num_outputs"])).to(device)
    stepper = sq_train.FusedTrainStep(model, lr=1e-3, world_size=world)

    def step():
        stepper.step(x, y)

    def cpu_baseline():
        from oracle import vis_oracle
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        nb = 8
        xc, yc = x[:nb].cpu(), y[:nb].cpu()
        m = {k: torch.zeros_like(v) for k, v in sd.items()}
        v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
        state = {"step": 0}

        def call():
            _, _, grads = vis_oracle.vis_loss_and_grads(sd, xc, yc)
            state["step"] += 1
            vis_oracle.adamw_step(sd, grads, m, v2, state["step"])
        rate, threads, reps = timed_cpu_sample(call, nb)
        return {"value": round(rate, 3), "unit": "slides/s", "cores": threads, "kind": "port",
                "sample": f"oracle ViS fwd+bwd+AdamW (torch-CPU fp32 autograd), {reps} steps x batch {nb}"}

    return dict(step=step, slides_per_step=B, cpu_baseline=cpu_baseline,
                config={"workload": "vis_train: ViS(D=1024 UNI-dim, depth 6, 16 heads, G=20820) fwd+bwd+AdamW on "
                                    "100 cluster tokens/slide (BASELINE config 2)",
                        "batch_per_gpu": B, "global_batch": B * world,
                        "parallelism": f"dp{world}" + (" (RCCL all-reduce of the flat gradient)" if world > 1 else "")})


def workload_pipeline(args, rank, world, device):
    """BASELINE config 3: 1000 x 224 x 224 uint8 patches/slide -> ResNet-50 -> k-Means(100) -> ViS forward."""
    from sequoia_pub_amd.pipeline import SlidePipeline
    from sequoia_pub_amd.resnet import resnet50
    from sequoia_pub_amd.vis import ViS
    nslides, npatch = args.slides, args.patches
    torch.manual_seed(99)
    uni = args.embedder == "uni"
    if uni:                                    # the UNI ViT-L/16 extractor (compute_features_hdf5.py --feat_type uni): 1024-d features
        from sequoia_pub_amd.uni import create_model
        rn = create_model(compute_dtype=args.dtype).to(device).eval()
        with torch.no_grad():                  # LayerScale at its init value (1e-5) would switch the blocks off numerically
            for k, (off, shape) in rn._tmap.items():
                if k.endswith("gamma"):
                    rn.flat[off:off + shape[0]] = 0.3
    else:
        rn = resnet50(pretrained=False, compute_dtype=args.dtype).to(device).eval()      # bf16x3: split-bf16 embedder (parity-grade)
        for m in rn.modules():                 # non-trivial BN statistics so folding is exercised
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    cfg = dict(VIS_CFG, input_dim=1024 if uni else 2048)
    vis_dtype = "fp32" if args.dtype in ("bf16x3", "f16x3") else args.dtype      # the aggregator (0.2 % of the FLOP) stays exact fp32 in the split mode
    vis = ViS(**cfg, num_clusters=100, device=str(device), compute_dtype=vis_dtype).to(device).eval()
    pipe = SlidePipeline(rn, vis, sub_batch=min(args.sub_batch, (args.uni_sub_batch or (1000 if args.dtype == "bf16" else 256))) if uni else args.sub_batch)
    # streaming form (default): the last slide's k-Means + ViS forward of a step run under the next step's first ResNet;
    # whatever is still in flight is flushed inside the timed region.  --no-stream: every step completes on its own.
    run = pipe if args.no_stream else pipe.submit
    S = args.patch_size
    host = [torch.from_numpy(synth.patches_u8(rank * nslides + i, npatch, S)) for i in range(nslides)]
    if args.from_host:
        # BASELINE config 3 as stated (the default): slides live in pinned host memory; each step uploads them on a
        # copy stream, slide i+1's upload under slide i's embedding
        host = [h.pin_memory() for h in host]
        copy_stream = torch.cuda.Stream(device=device)
        # two sets of staging buffers: step s uploads into set s % 2 while step s-1's slides are still being embedded; a
        # set is free again once the step that used it two steps ago has been consumed by the main stream
        staging = [[torch.empty_like(h, device=device) for h in host] for _ in range(2)]
        consumed = [None, None]
        state = {"step": 0}

        def step():
            main = torch.cuda.current_stream(device)
            k = state["step"] & 1
            state["step"] += 1
            if consumed[k] is not None:
                copy_stream.wait_event(consumed[k])          # the main stream is done with this set's previous contents
            evs = []
            with torch.cuda.stream(copy_stream):
                for h, d in zip(host, staging[k]):
                    d.copy_(h, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                    evs.append(ev)
            run(list(zip(staging[k], evs)))      # the pipeline waits for slide i's upload right before embedding it
            done = torch.cuda.Event()
            done.record(main)                    # (streaming form: the set's last slide may still be in flight -- see flush)
            consumed[k] = done
    else:
        slides = [h.to(device) for h in host]

        def step():
            run(slides)

    def cpu_baseline():
        """BASELINE config 1 on this host's cores, bounded: the literal batch-1 patch loop of
        compute_features_hdf5.py:116-123 (one forward + one result copy per patch) on a sample of patches, the same
        sample batched (so the speed-up is not inflated by the reference's loop structure), then ONE oracle
        k-Means(100) + cluster means + ViS forward; per-slide time = 1000 x per-patch time + the rest."""
        from oracle import kmeans_oracle, resnet_oracle, uni_oracle, vis_oracle
        sd_r = {k: v.cpu() for k, v in rn.state_dict().items()}
        sd_v = {k: v.cpu() for k, v in vis.state_dict().items()}
        ncpu = os.cpu_count() or 1
        torch.set_num_threads(min(ncpu, 64))
        n_lit, n_bat = (12, 16) if uni else (96, 128)
        embed = (lambda p, batch: uni_oracle.embed_patches(sd_r, p, heads=16, batch=batch)) if uni else \
                (lambda p, batch: resnet_oracle.embed_patches(sd_r, p, batch=batch))
        sample = host[0][:n_bat].cpu() if isinstance(host[0], torch.Tensor) else host[0][:n_bat]
        embed(sample[:2], 1)                                                        # warm the thread pool
        t0 = time.perf_counter()
        out = [embed(sample[i:i + 1], 1)[0].numpy() for i in range(n_lit)]
        t_lit = (time.perf_counter() - t0) / n_lit
        t0 = time.perf_counter()
        embed(sample, n_bat)
        t_bat = (time.perf_counter() - t0) / n_bat
        feats = synth.features_gmm(5, npatch, 1024 if uni else 2048)
        t0 = time.perf_counter()
        r = kmeans_oracle.kmeans_fit(feats)
        cf = kmeans_oracle.cluster_means(feats, r["labels"])
        with torch.no_grad():
            vis_oracle.vis_forward(sd_v, torch.from_numpy(cf)[None])
        t_rest = time.perf_counter() - t0
        lit, bat = 1.0 / (npatch * t_lit + t_rest), 1.0 / (npatch * t_bat + t_rest)
        return {"value": round(lit, 5), "unit": "slides/s", "cores": torch.get_num_threads(), "kind": "port",
                "batched_value": round(bat, 5),
                "sample": f"config 1 restated (oracle/, torch-CPU fp32): literal batch-1 loop on {n_lit} patches "
                          f"({t_lit * 1e3:.1f} ms/patch) and the same arithmetic in one batch of {n_bat} ({t_bat * 1e3:.1f} ms/patch), "
                          f"each extrapolated to {npatch} patches, + one oracle k-Means(100) + cluster means + ViS forward "
                          f"({t_rest:.2f} s); value = literal, batched_value = batched"}

    return dict(step=step, flush=None if args.no_stream else pipe.flush, slides_per_step=nslides, cpu_baseline=cpu_baseline,
                config={"workload": f"pipeline: {npatch} x {S}x{S} uint8 patches/slide -> " +
                                    ("UNI ViT-L/16 embed -> k-Means(100) -> ViS(D=1024, depth 6, 16 heads, G=20820) forward (the metric's UNI-dim variant), "
                                     if uni else "ResNet-50 embed -> k-Means(100) -> ViS(D=2048, depth 6, 16 heads, G=20820) forward (BASELINE config 3), ") +
                                    ("patches uploaded from pinned host memory every step" if args.from_host else "patches resident in HBM"),
                        "embedder": "UNI ViT-L/16 (parity unpinned: timm + gated weights absent)" if uni else "ResNet-50 (src/resnet.py forward_extract)",
                        "feature_dim": 1024 if uni else 2048, "patch_size": S,
                        "slides_per_step_per_gpu": nslides, "patches_per_slide": npatch,
                        "parallelism": f"slide-sharded x{world}"})


def workload_spatial(args, rank, world, device):
    """BASELINE config 5: one slide of 50 000 tiles on a 250 x 200 grid, 10 x 10 windows at stride 1 (no k-Means),
    ViS forward per window, per-tile mean of the 20 820-gene predictions.  With --gpus N > 1 the ONE slide is dealt over the
    ranks (window batches and tile chunks, spatial.sliding_window_all_genes_sharded; one all-gather of the window vectors per
    slide): total work is fixed, so the line says "scaling": "strong" and `value` is slides per second of the whole job."""
    from sequoia_pub_amd.spatial import enumerate_windows, sliding_window_all_genes_sharded
    from sequoia_pub_amd.vis import ViS
    torch.manual_seed(99)
    nx, ny = args.grid
    xs, ys = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    xtf, ytf = xs.ravel(), ys.ravel()
    vis = ViS(**VIS_CFG, num_clusters=100, device=str(device), compute_dtype=args.dtype).to(device).eval()
    feats = torch.randn(nx * ny, VIS_CFG["input_dim"], generator=torch.Generator().manual_seed(99)).to(device)      # the same slide on every rank
    n_windows = len(enumerate_windows(xtf, ytf, 1)[0])
    shard = (rank, world) if world > 1 else None
    bw = args.batch_windows
    if world > 1:                           # at least two batches per rank, so that the two window streams of a rank both have work
        bw = min(bw, max(256, -(-n_windows // (2 * world) // 256) * 256))

    def step():
        out, _, _ = sliding_window_all_genes_sharded(xtf, ytf, feats, vis, 1, batch_windows=bw, shard=shard)
        return out

    def cpu_baseline():
        from oracle import vis_oracle
        sd = {k: v.cpu() for k, v in vis.state_dict().items()}
        x = torch.randn(16, 100, VIS_CFG["input_dim"])

        def fwd():
            with torch.no_grad():
                vis_oracle.vis_forward(sd, x)
        rate, threads, reps = timed_cpu_sample(fwd, 16, budget_s=10.0)
        return {"value": round(rate / n_windows, 6), "unit": "slides/s", "cores": threads, "kind": "port",
                "sample": f"oracle ViS forward on {reps} x 16 windows ({rate:.1f} windows/s) extrapolated to {n_windows} windows; "
                          "feature cache assumed (the reference re-embeds every tile per window), voting not timed"}

    return dict(step=step, slides_per_step=1.0 / world, cpu_baseline=cpu_baseline, scaling="strong" if world > 1 else "weak",
                config={"workload": f"spatial: {nx * ny} tiles ({nx} x {ny} grid), {n_windows} windows of 100 tokens at stride 1, "
                                    "ViS(D=1024, depth 6, 16 heads, G=20820) forward per window + per-tile mean vote (BASELINE config 5)",
                        "windows_per_slide": n_windows, "batch_windows": bw,
                        "parallelism": f"window-sharded x{world}" if world > 1 else "one slide on one GPU"})


def workload_train_kfold(args, rank, world, device):
    """BASELINE config 4: src/main.py:101-219 semantics on synthetic slides -- 64 slides per GPU (512 on 8 GPUs), one
    slide per synthetic patient, patient_kfold 5 folds; per fold a freshly initialised ViS is trained for
    --epochs epochs (train + val phases through train(): fused step, device-side metrics, RCCL gradient all-reduce
    under DDP, save / stop policy without checkpoint files) and evaluated on the fold's test part.
    A step = one full 5-fold pass; slides/s counts every slide forward (train, val and test)."""
    import pandas as pd
    from sequoia_pub_amd import train as sq_train
    from sequoia_pub_amd.data import patient_kfold
    per_gpu, E = args.batch, args.epochs
    n = per_gpu * world
    model = make_vis(args.dtype, device)
    init = model.flat.detach().clone()
    df = pd.DataFrame(dict(patient_id=[f"P{i:05d}" for i in range(n)]))
    folds = list(zip(*patient_kfold(df, n_splits=5)))
    # Every rank holds the whole synthetic cohort on its device (512 slides = 250 MB: nothing at 288 GB); a loader batch is
    # a GLOBAL batch of per_gpu * world consecutive rows of the fold's list, of which rank r takes the r-th contiguous
    # chunk -- so the global batches, and with them the trajectory, do not depend on the number of ranks (a ragged last
    # batch leaves the last ranks short or empty: the exchange step handles that).
    x_all = torch.from_numpy(synth.cluster_tokens(99, n, 1024)).to(device)
    y_all = torch.from_numpy(synth.rna_targets(199, n, VIS_CFG["num_outputs"])).to(device)
    gb = per_gpu * world

    def loader(rows):
        rows = np.asarray(rows)
        out = []
        for b0 in range(0, len(rows), gb):
            mine = rows[b0 + rank * per_gpu: min(b0 + (rank + 1) * per_gpu, b0 + gb, len(rows))]
            if len(mine) == 0:
                out.append(([], [], [], []))
                continue
            sel = torch.as_tensor(mine, dtype=torch.long, device=device)
            out.append((x_all[sel], y_all[sel], [f"slide{int(r)}" for r in mine], ["SYN"] * len(mine)))
        return out

    state = {"folds_run": 0}

    def step():
        state["folds_run"] = 0
        for i, (tr, va, te) in enumerate(folds):
            with torch.no_grad():
                model.flat.copy_(init)                       # main.py:165: a new model per fold (bumps the version: bf16 shadow refreshed)
            sq_train.train(model, {"train": loader(tr), "val": loader(va)}, None, num_epochs=E, save_dir=None,
                           verbose=False, split=i, lr=1e-3, grad_exchange="bf16" if args.dtype == "bf16" else "fp32")     # config 4 states bf16
            sq_train.evaluate(model, loader(te), verbose=False)
            state["folds_run"] += 1

    def check():
        """After the timed region: what a reader (and tests/test_gpu_ddp.py) needs to see that the ranks did one job --
        folds completed, the parameters of the last fold identical on every rank, their sums for a cross-run comparison."""
        flat = model.flat.detach()
        sums = torch.stack([flat.double().sum(), flat.double().abs().sum()])
        same = True
        if world > 1:
            allp = [torch.empty_like(sums) for _ in range(world)]
            torch.distributed.all_gather(allp, sums)
            same = all(bool(torch.equal(a, allp[0])) for a in allp)
        dump = os.environ.get("SQ_BENCH_DUMP_PARAMS")
        if dump and rank == 0:
            torch.save(flat.cpu(), dump)
        return {"folds_run": state["folds_run"], "ranks_hold_identical_parameters": same,
                "param_sum": float(sums[0]), "param_abs_sum": float(sums[1]),
                # per training step of the last epoch (rank 0): backward pass, bucketed all-reduce busy time on the communication
                # stream, and the part of it that stuck out behind the backward pass -- None on one rank
                "gradient_exchange_ms_per_step": getattr(model, "last_exchange_timing", None)}

    def cpu_baseline():
        from oracle import vis_oracle
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        nb = 8
        xc, yc = x_all[:nb].cpu(), y_all[:nb].cpu()
        m = {k: torch.zeros_like(v) for k, v in sd.items()}
        v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
        state = {"step": 0}

        def call():
            _, _, grads = vis_oracle.vis_loss_and_grads(sd, xc, yc)
            state["step"] += 1
            vis_oracle.adamw_step(sd, grads, m, v2, state["step"])
        rate, threads, reps = timed_cpu_sample(call, nb)
        return {"value": round(rate, 3), "unit": "slides/s", "cores": threads, "kind": "port",
                "sample": f"oracle ViS fwd+bwd+AdamW (torch-CPU fp32 autograd), {reps} steps x batch {nb}; host-side metrics not included"}

    return dict(step=step, slides_per_step=per_gpu * (4 * E + 1), cpu_baseline=cpu_baseline, check=check,
                config={"workload": f"train_kfold: {n} synthetic slides ({per_gpu}/GPU), patient_kfold 5 folds, {E} epochs per fold "
                                    "(train + val) + test evaluation, ViS(D=1024, depth 6, 16 heads, G=20820) (BASELINE config 4)",
                        "slides": n, "epochs_per_fold": E,
                        "parallelism": f"dp{world}" + (" (RCCL all-reduce of the flat gradient, bucketed under the backward pass)" if world > 1 else "")})


GOLDEN_SLIDES = {        # reference-made slides (tests/golden/make_golden.py gold_pipeline, gold_pipeline_hard): fixture, patches, weight set
    "noise224": ("pipeline_slide.npz", lambda: synth.patches_u8(7, 1000, 224), "std"),
    "struct224": ("pipeline_slide_struct224.npz", lambda: synth.structured_patches_u8(11, 1000, 224), "std"),
    "struct256": ("pipeline_slide_struct256.npz", lambda: synth.structured_patches_u8(12, 1000, 256), "std"),
    "wide224": ("pipeline_slide_wide224.npz", lambda: synth.structured_patches_u8(13, 1000, 224), "wide"),
}


def accuracy_vs_reference(dtype_name, device, slides=("noise224",)):
    """Checker leg (rank 0, N = 1, outside every timed region): 1000-patch slides made by the REFERENCE's resnet50 +
    scikit-learn KMeans + ViS (tests/golden/make_golden.py) through SlidePipeline in the mode the line is quoted in --
    the uniform-noise slide, structured patches (white background, saturated / near-black / flat regions) at 224 and 256 px,
    and a weight set whose BN statistics span > 4 decades.  What a reader needs to price the throughput number: feature
    error (max-norm and the share of elements within 1e-4 in the allclose form), how many of the 1000 cluster labels equal
    scikit-learn's, partition agreement, 20 820-gene prediction error, slides that had to be re-run in exact fp32.
    oracle/ supplies only the seeded weight recipes of the goldens (no arithmetic of the path)."""
    from oracle import resnet_oracle as ro, vis_oracle
    from sequoia_pub_amd.pipeline import SlidePipeline
    from sequoia_pub_amd.resnet import resnet50
    from sequoia_pub_amd.vis import ViS
    gdir = os.path.join(ROOT, "tests", "golden")
    cfg = dict(VIS_CFG, input_dim=2048)
    vis = ViS(**cfg, device=str(device), compute_dtype="fp32" if dtype_name in ("bf16x3", "f16x3") else dtype_name)
    vis.load_state_dict(vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=31), seed=32))
    vis = vis.to(device).eval()

    def rel(a, b):
        return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())

    def frac_close(a, b, rtol):
        b = np.asarray(b, np.float64)
        return float((np.abs(np.asarray(a, np.float64) - b) <= rtol * (np.abs(b) + np.abs(b).max())).mean())

    res = {"mode": dtype_name, "golden": "tests/golden/pipeline_slide*.npz (reference resnet50 + scikit-learn KMeans(100, random_state=0) + ViS, fp32 CPU)",
           "kmeans_caveat": "labels_equal is against scikit-learn on these committed slides; the deterministic k-Means definition (fp64-summed k-means++ "
                            "potentials) parts from scikit-learn's fp32-BLAS-dependent result on 4 of 132 and 6 of 256 scanned slides "
                            "(tests/golden/kmeans_sklearn_mismatch.json, kmeans_sklearn_scan_goldenlike.json)"}
    nets = {}
    for key in slides:
        t_slide = time.perf_counter()
        fixture, make_patches, weights = GOLDEN_SLIDES[key]
        path = os.path.join(gdir, fixture)
        if not os.path.exists(path):
            continue
        z = np.load(path)
        if weights not in nets:
            sd = ro.init_resnet50_state_dict(seed=99, perturb_bn=True) if weights == "std" else \
                ro.init_resnet50_state_dict_wide(123, running_stats=np.load(os.path.join(gdir, "resnet50_wide_bn.npz")))
            rn = resnet50(pretrained=False, compute_dtype=dtype_name)
            full = rn.state_dict()
            full.update(sd)
            rn.load_state_dict(full)
            nets[weights] = rn.to(device).eval()
        t_net = time.perf_counter() - t_slide
        pipe = SlidePipeline(nets[weights], vis, n_clusters=100, sub_batch=500)
        t0 = time.perf_counter()
        patches = torch.from_numpy(make_patches())
        t_gen = time.perf_counter() - t0
        if os.environ.get("SQ_BENCH_ACC_TRACE"):       # debugging aid: the stages one by one, synchronised
            pd = patches.to(device)
            for name, fn in (("upload sync", lambda: None), ("embed", lambda: pipe.embed(pd)), ("cluster", lambda: pipe.cluster(pipe.embed(pd).unsqueeze(0))),
                             ("vis", lambda: vis(torch.zeros(1, 100, 2048, device=device)))):
                t1 = time.perf_counter(); fn(); torch.cuda.synchronize()
                print(f"accuracy trace {key}: {name} {time.perf_counter() - t1:.2f} s", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        out = pipe([patches.to(device)])
        torch.cuda.synchronize()
        t_run = time.perf_counter() - t0
        feats = out["features"][0].cpu().numpy()
        labels = out["labels"][0].cpu().numpy().astype(np.int64)
        pred = out["pred"][0].cpu().numpy().astype(np.float64)
        ref_l = z["labels"].astype(np.int64)
        step = int(z["probe_step"]) if "probe_step" in z else 64
        cont = np.zeros((100, 100), dtype=np.int64)
        np.add.at(cont, (labels, ref_l), 1)
        comb = lambda x: int((x * (x - 1) // 2).sum())
        total = 1000 * 999 // 2
        rand = (total + 2 * comb(cont) - comb(cont.sum(1)) - comb(cont.sum(0))) / total
        res[key] = {"feature_rel_err": float(f"{rel(feats[::step], z['feat_probe'].astype(np.float64)):.3e}"),
                    "feature_allclose_1e-4_fraction": round(frac_close(feats[::step], z["feat_probe"], 1e-4), 6),
                    "labels_equal": int((labels == ref_l).sum()), "labels_total": 1000,
                    "kmeans_n_iter_reference": int(z["n_iter"]),
                    "partition_rand_index": round(float(rand), 6),
                    "prediction_rel_err": float(f"{rel(pred, z['pred'].astype(np.float64)):.3e}"),
                    "slides_rerun_in_fp32": int(getattr(pipe, "nonfinite_reruns", 0)),
                    "checker_seconds": {"weights": round(t_net, 1), "patches": round(t_gen, 1), "pipeline": round(t_run, 1),
                                        "total": round(time.perf_counter() - t_slide, 1)}}
    return res


WORKLOADS = {"vis_fwd": workload_vis_fwd, "vis_train": workload_vis_train, "pipeline": workload_pipeline, "spatial": workload_spatial,
             "train_kfold": workload_train_kfold}


METRIC = "slides/sec (1000-patch WSI, UNI-dim, 20k-gene head)"
RESNET_FLOP_PER_PATCH = 8.174e9          # 4.087 GMAC per 224 x 224 patch
UNI_FLOP_PER_PATCH = 122.6e9             # ViT-L/16 at 197 tokens: 61.3 GMAC (24 blocks x (4 D^2 + 2 D mlp) per token + attention)
VIS_FWD_FLOP = {1024: 5.17e9, 2048: 15.4e9}   # per slide, algorithmic (s(mean x) shortcut): 6 layers of f / projection / 2 FF products + head


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks ourselves (one process per GPU, RCCL rendezvous on 127.0.0.1)."""
    import subprocess
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def measure(name, args, rank, world, device, want_roofline=True, want_cpu=True):
    """Build workload `name`, time it (barrier + synchronize on both sides, max over ranks), optionally profile the
    dominant kernel and the CPU baseline.  Returns the fields of a bench line for this workload."""
    wl = WORKLOADS[name](args, rank, world, device)
    steps, warmup = args.steps, args.warmup
    if steps is None:
        # default step count: long enough that the timed region is >= ~2.5 s (a 70 ms region is not a measurement)
        for _ in range(max(warmup, 1)):
            wl["step"]()
        if wl.get("flush"):
            wl["flush"]()
        barrier_sync(world)
        t0 = time.perf_counter()
        wl["step"]()
        if wl.get("flush"):
            wl["flush"]()
        barrier_sync(world)
        est = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            torch.distributed.all_reduce(est, op=torch.distributed.ReduceOp.MAX)
        steps = int(min(2000, max(5, math.ceil(2.5 / max(float(est.item()), 1e-4)))))
        warmup = 0 if warmup is None else max(0, warmup - 1)
    dt = timed_region(wl["step"], steps, warmup, world, device, wl.get("flush"))
    value = wl["slides_per_step"] * world * steps / dt
    out = {"value": round(value, 3), "unit": "slides/s", "steps": steps, "warmup": args.warmup,
           "ms_per_step": round(dt / steps * 1e3, 4), "timed_region_s": round(dt, 3), "dtype": args.dtype, "config": wl["config"],
           "scaling": wl.get("scaling", "weak")}
    if LAST_POWER[0]:
        out["power"] = LAST_POWER[0]
        LAST_POWER[0] = None
        if out["power"].get("package_w") and value > 0:
            # the headline workload runs at the package power cap: joules per slide is what a change has to lower
            out["power"]["energy_j_per_slide"] = round(out["power"]["package_w"] * world / value, 3)
    recs = []
    if want_roofline:
        roof, recs = roofline_from_profile(wl["step"], min(steps, 3), args.dtype, name)
        if roof is not None and out.get("power") and roof.get("bound") == "mfma":
            # the same fraction against the matrix peak AT THE CLOCK THE CHIP SUSTAINED in the timed region
            roof["frac_at_sustained_clock"] = round(roof["frac"] / max(out["power"]["sclk_frac_of_max"], 1e-3), 4)
        if wl.get("flush"):
            wl["flush"]()
        if roof is not None and name == "pipeline":
            flop = args.patches * (UNI_FLOP_PER_PATCH if args.embedder == "uni" else RESNET_FLOP_PER_PATCH * (args.patch_size / 224.0) ** 2) + VIS_FWD_FLOP[1024 if args.embedder == "uni" else 2048]
            roof["end_to_end"] = {"algorithmic_tflop_per_slide": round(flop / 1e12, 3),
                                  "achieved_tflops": round(flop * value / world / 1e12, 1), "peak_tflops": PEAK[args.dtype],
                                  "frac_of_mfma_peak": round(flop * value / world / 1e12 / PEAK[args.dtype], 4)}
        if roof is not None and name == "spatial":
            # BASELINE config 5 asks for the HBM view: the per-tile prediction matrix (tiles x genes, fp32) is the floor
            # of what must reach HBM; the per-window predictions (windows x genes) are voted before the head and never
            # materialised.  The workload is matrix-bound: both fractions are reported.
            nw = wl["config"]["windows_per_slide"]
            flop = nw * VIS_FWD_FLOP[1024]
            out_bytes = args.grid[0] * args.grid[1] * 20820 * 4
            roof["end_to_end"] = {"algorithmic_tflop_per_slide": round(flop / 1e12, 2),
                                  "achieved_tflops": round(flop * value / world / 1e12, 1), "peak_tflops": PEAK[args.dtype],
                                  "frac_of_mfma_peak": round(flop * value / world / 1e12 / PEAK[args.dtype], 4),
                                  "output_floor_bytes_per_slide": out_bytes,
                                  "output_floor_gbs": round(out_bytes * value / world / 1e9, 1), "hbm_peak_gbs": HBM_PEAK_GBS,
                                  "frac_of_hbm_peak_at_output_floor": round(out_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5)}
        if roof is not None and name in ("vis_train", "vis_fwd"):
            flop = args.batch * VIS_FWD_FLOP[1024] * (3 if name == "vis_train" else 1)          # fwd + bwd = 3 x the algorithmic forward
            roof["end_to_end"] = {"algorithmic_tflop_per_step": round(flop / 1e12, 4), "achieved_tflops": round(flop * value / world / args.batch / 1e12, 1),
                                  "peak_tflops": PEAK[args.dtype], "frac_of_mfma_peak": round(flop * value / world / args.batch / 1e12 / PEAK[args.dtype], 4)}
        if roof is not None and roof.get("end_to_end"):
            # the dominant class is one kernel of many (11 % of the headline step): the whole step against the same peak
            roof["frac_end_to_end"] = roof["end_to_end"]["frac_of_mfma_peak"]
        out["roofline"] = roof
    if want_cpu and rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = wl["cpu_baseline"]()
    out["_recs"] = recs
    if wl.get("check"):
        out["check"] = wl["check"]()
    del wl
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: as many as make the timed region >= 2.5 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("SQ_BENCH_WORKLOAD", "pipeline"), choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32", "bf16x3", "f16x3"],
                    help="default: f16x3 (parity-grade split fp16) for the pipeline with the ResNet-50 embedder, bf16 otherwise")
    ap.add_argument("--batch", type=int, default=64, help="slides per GPU per step (vis_* / train_kfold)")
    ap.add_argument("--epochs", type=int, default=2, help="train_kfold workload: epochs per fold")
    ap.add_argument("--slides", type=int, default=8, help="pipeline workload: slides per GPU per step")
    ap.add_argument("--patches", type=int, default=1000, help="pipeline workload: patches per slide")
    ap.add_argument("--patch-size", type=int, default=224, help="pipeline workload: patch edge in pixels (224 = BASELINE config 3; 256 = the reference's default, patch_gen_hdf5.py:157)")
    ap.add_argument("--sub-batch", type=int, default=1000, help="pipeline workload: patches per ResNet launch group (two groups in flight)")
    ap.add_argument("--uni-sub-batch", type=int, default=0, help="pipeline workload, UNI embedder: patches per launch group (default 1000 in bf16 "
                    "-- 6.09 / 6.38 / 6.58 slides/s at 256 / 500 / 1000 -- and 256 in fp32: the 2 GiB descriptor limit)")
    ap.add_argument("--grid", type=int, nargs=2, default=[250, 200], help="spatial workload: tile grid")
    ap.add_argument("--batch-windows", type=int, default=8192, help="spatial workload: windows per ViS forward (8192: 438 ms per slide against 444 at 2048, 470 at 1024 and 505 at 512 -- more tiles per launch of the 256 x 256 GEMM)")
    ap.add_argument("--embedder", default="resnet", choices=["resnet", "uni"], help="pipeline workload: patch embedder")
    ap.add_argument("--no-stream", action="store_true", help="pipeline workload: finish every step's slides before the next step starts")
    ap.add_argument("--resident", action="store_true", help="pipeline workload: patches already in HBM when the timed region starts "
                    "(default: uploaded from pinned host memory every step, as BASELINE config 3 states)")
    ap.add_argument("--measure-traffic", action="store_true", help="collect roofline.traffic with two rocprofv3 --pmc passes of this build "
                    "(the default for the default workload when rocprofv3 is on the box)")
    ap.add_argument("--no-measure-traffic", action="store_true", help="keep the figure of the committed pass under profiles/ (traffic_measured false)")
    ap.add_argument("--no-accuracy", action="store_true", help="pipeline workload: skip the accuracy_vs_reference checker leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="default workload only: skip the secondary measurements")
    args = ap.parse_args()
    args.from_host = not args.resident
    if args.dtype is None:
        args.dtype = "f16x3" if (args.workload == "pipeline" and args.embedder == "resnet") else "bf16"
    if args.dtype in ("bf16x3", "f16x3") and not (args.workload == "pipeline" and args.embedder == "resnet"):
        raise SystemExit(f"--dtype {args.dtype} is the split mode of the ResNet-50 embedder (pipeline workload)")

    share = os.environ.get("SQ_BENCH_SHARE_GPU") == "1"
    if args.gpus > 1 and not share:
        have = torch.cuda.device_count()
        if have < args.gpus:          # said here, once, instead of N ranks fighting over cuda:0 or hanging in the RCCL rendezvous
            raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this box has {have} "
                             f"(one rank per GPU over RCCL; SQ_BENCH_SHARE_GPU=1 runs the ranks on cuda:0 over gloo to exercise the control flow only)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    rank, world, local = dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself when no launcher is present)")
    _lib.require_gpu()
    # SQ_BENCH_SHARE_GPU=1 (debugging aid for 1-GPU boxes): every rank uses cuda:0 and the collectives go through
    # gloo -- exercises the multi-process control flow (rendezvous, bucketed all-reduce, barriers), not RCCL
    device = torch.device("cuda", 0 if share else local)
    torch.cuda.set_device(device)
    from sequoia_pub_amd.cli.common import bind_to_gpu_numa_node
    numa = bind_to_gpu_numa_node(device.index)          # host thread + later pinned allocations next to this rank's GPU
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)
        got = torch.distributed.get_world_size()
        if got != args.gpus:
            raise SystemExit(f"process group has {got} ranks, --gpus asked for {args.gpus}")
        if rank == 0:
            print(f"bench: {got} ranks joined the {'gloo (shared GPU)' if share else 'RCCL'} process group", file=sys.stderr)

    res = measure(args.workload, args, rank, world, device)
    line = {"metric": METRIC, "value": res["value"], "unit": "slides/s", "n_gpus": world, "steps": res["steps"],
            "warmup": res["warmup"], "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": res.get("scaling", "weak"),
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "config": res["config"],
            "roofline": res.get("roofline"), "cpu_baseline": res.get("cpu_baseline"), "timed_region_s": res["timed_region_s"],
            "host_numa_binding": numa, "power": res.get("power"),
            "ranks": torch.distributed.get_world_size() if world > 1 else 1,
            "backend": (torch.distributed.get_backend() if world > 1 else None)}
    default_wl = args.workload == "pipeline" and args.dtype == "f16x3" and args.embedder == "resnet"
    if (args.measure_traffic or (default_wl and not args.no_measure_traffic)) and rank == 0 and world == 1 and line.get("roofline"):
        try:
            (measure_traffic if default_wl else measure_traffic_marked)(line["roofline"], args, res.get("_recs") or ())
        except Exception as e:                          # a profiler problem must not cost the line
            line["roofline"]["traffic_error"] = f"{type(e).__name__}: {e}"
    if "check" in res:
        line["check"] = res["check"]
    if args.workload == "pipeline" and args.embedder == "resnet" and rank == 0 and world == 1 and not args.no_accuracy:
        try:
            line["accuracy_vs_reference"] = accuracy_vs_reference(args.dtype, device, tuple(GOLDEN_SLIDES) if args.dtype == "f16x3" else ("noise224",))
        except Exception as e:                          # the checker leg must not cost the line
            line["accuracy_vs_reference"] = {"error": f"{type(e).__name__}: {e}"}

    # secondary measurements of the default run (N = 1): the PCIe-inclusive twin, BASELINE config 2, the fp32 parity
    # mode.  Each runs in a fresh process (same script, same timing rules): a process that has already created one
    # workload's helper streams maps later streams onto the same few hardware queues, which serialises what the
    # next workload wants to overlap (measured: from-host 41 instead of 53 slides/s, vis_train 10.5 instead of 3.6 ms).
    if args.workload == "pipeline" and args.dtype == "f16x3" and args.from_host and args.embedder == "resnet" and world == 1 and not args.no_secondary:
        import subprocess
        sec = {}
        for key, extra in (("pipeline_resident_in_hbm", ["--workload", "pipeline", "--resident", "--no-accuracy"]),
                           ("pipeline_bf16_throughput_mode_from_pinned_host", ["--workload", "pipeline", "--dtype", "bf16"]),
                           ("pipeline_fp32_exact_mfma_mode", ["--workload", "pipeline", "--dtype", "fp32", "--slides", "2", "--sub-batch", "250", "--no-accuracy"]),
                           ("pipeline_256px_patches", ["--workload", "pipeline", "--patch-size", "256", "--slides", "4", "--no-accuracy"]),
                           ("vis_train_bf16", ["--workload", "vis_train"]),
                           ("train_kfold_64_slides_per_gpu", ["--workload", "train_kfold"]),      # BASELINE config 4's per-GPU share on this one GPU
                           ("pipeline_uni_vit_l16_embedder", ["--workload", "pipeline", "--embedder", "uni", "--slides", "2"]),
                           ("spatial_50k_tiles", ["--workload", "spatial"])):
            in_run = key in ("vis_train_bf16", "pipeline_uni_vit_l16_embedder", "spatial_50k_tiles")      # one marker-attributed --pmc pass pair each
            cmd = [sys.executable, os.path.abspath(__file__), "--no-secondary", "--no-cpu-baseline", "--measure-traffic" if in_run else "--no-measure-traffic",
                   "--warmup", str(args.warmup)] + extra
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                last = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not last:
                    raise RuntimeError((r.stderr or r.stdout)[-400:])
                d = json.loads(last[-1])
                sec[key] = {k: d.get(k) for k in ("value", "unit", "steps", "warmup", "ms_per_step", "timed_region_s", "dtype", "config", "roofline", "power",
                                              "accuracy_vs_reference") if k in d}
            except Exception as e:                      # a secondary failure must not cost the headline line
                sec[key] = {"error": f"{type(e).__name__}: {e}"}
        line["secondary"] = sec
        fp32 = sec.get("pipeline_fp32_exact_mfma_mode", {})
        if "value" in fp32:         # the same pipeline in the reference's own arithmetic (exact fp32 FMA chains on v_mfma_f32_32x32x2_f32)
            line["reference_arithmetic"] = {"value": fp32["value"], "unit": "slides/s", "dtype": "fp32"}

    if rank == 0:
        if os.environ.get("SQ_BENCH_KERNELS"):
            with open(os.environ["SQ_BENCH_KERNELS"], "w") as f:
                json.dump(sorted(res["_recs"], key=lambda r: -r["total_ms"]), f, indent=1)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
