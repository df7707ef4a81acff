"""Worker of tests/test_gpu_ddp.py: one of two ranks that SHARE cuda:0 (a 1-GPU box), process group over gloo.
Drives FusedTrainStep(world_size=2) -- bucketed gradient exchange on the communication stream, ragged and empty
local batches -- and lets rank 0 compare with the one-rank step on the concatenated batch."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import sequoia_pub_amd  # noqa: E402,F401
from sequoia_pub_amd import train as sq_train  # noqa: E402
from sequoia_pub_amd.vis import ViS  # noqa: E402

CFG = dict(num_outputs=50, input_dim=128, depth=2, nheads=2, dimensions_f=64, dimensions_s=64, dimensions_c=64)


def model():
    torch.manual_seed(7)
    return ViS(**CFG, device="cuda:0", compute_dtype="fp32").to("cuda:0")


def trajectory(rank, world, ok_path, steps=12):
    """`world` ranks (8 in the test), two slides each, `steps` optimizer steps: the bf16 wire format against the fp32 wire format
    and against ONE rank on the concatenated batch -- the full-batch loss after every step and the parameters at the end.  Rounding
    on the wire grows with the ring length (world - 1 partial sums), which two ranks for two steps do not show."""
    g = torch.Generator().manual_seed(11)
    n = 2 * world
    x = torch.randn(n, 100, 128, generator=g).cuda()
    y = (torch.rand(n, 50, generator=g) * 8).cuda()
    rows = slice(2 * rank, 2 * rank + 2)

    def run(wire, ranks):
        m = model()
        st = sq_train.FusedTrainStep(m, lr=1e-3, world_size=ranks, grad_exchange=wire)
        losses = []
        for _ in range(steps):
            if ranks > 1:
                st.step(x[rows], y[rows], n_global=n * 50)
            else:
                st.step(x, y)
            with torch.no_grad():
                losses.append(float(((m(x) - y) ** 2).mean()))
        torch.cuda.synchronize()
        return losses, m.flat.detach().clone()

    lb, pb = run("bf16", world)
    lf, pf = run("fp32", world)
    for p in (pb, pf):                                   # whatever the wire format, every rank must hold the same parameters
        allp = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(allp, p)
        assert all(torch.equal(a, allp[0]) for a in allp), "ranks diverged"
    if rank == 0:
        l1, p1 = run("fp32", 1)
        gap_b = max(abs(a - b) / b for a, b in zip(lb, l1))
        gap_f = max(abs(a - b) / b for a, b in zip(lf, l1))
        db, df = (pb - p1).abs(), (pf - p1).abs()
        print(f"ddp{world} trajectory over {steps} steps: loss {l1[0]:.4f} -> {l1[-1]:.4f}; worst relative loss gap to the one-rank run: "
              f"bf16 wire {gap_b:.2e}, fp32 wire {gap_f:.2e}; parameter diff max/mean: bf16 {float(db.max()):.2e}/{float(db.mean()):.2e}, "
              f"fp32 {float(df.max()):.2e}/{float(df.mean()):.2e}")
        assert l1[-1] < l1[0]                              # it trains
        assert gap_f < 1e-4 and gap_b < 5e-3
        # AdamW at lr 1e-3: a parameter moves <= lr per step; a gradient whose sign the wire rounding flips costs <= 2 lr per step
        assert float(db.max()) <= 2e-3 * steps and float(db.mean()) < 2e-4 and float(df.mean()) < 2e-5
        open(ok_path, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    if len(sys.argv) > 2 and sys.argv[2] == "trajectory":
        return trajectory(rank, world, sys.argv[1])
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 100, 128, generator=g).cuda()
    y = (torch.rand(6, 50, generator=g) * 8).cuda()
    rows = slice(0, 4) if rank == 0 else slice(4, 6)            # ragged: 4 + 2 slides

    wire = sys.argv[2] if len(sys.argv) > 2 else "fp32"         # wire format of the gradient exchange: fp32 | bf16
    m = model()
    step = sq_train.FusedTrainStep(m, lr=1e-3, world_size=world, grad_exchange=wire)
    assert len(step.buckets) == CFG["depth"] + 1
    assert step.exchange_bytes_per_step == m.flat.numel() * (2 if wire == "bf16" else 4)
    out = step.step(x[rows], y[rows], n_global=6 * 50)          # step 1: both ranks hold slides
    assert out is not None
    g1 = m._gflat.clone()
    if rank == 0:                                               # step 2: rank 1's batch collated to nothing
        out = step.step(x[0:4], y[0:4], n_global=4 * 50)
    else:
        out = step.step(None, None, n_global=4 * 50)
        assert out is None
    torch.cuda.synchronize()
    g2 = m._gflat.clone()
    flat = m.flat.detach().clone()
    tr = step.timing_report()                                   # step 1 was sampled: backward time, all-reduce span, exposed part
    assert tr is not None and tr["sampled_steps"] == 1 and tr["backward_ms"] > 0 and tr["allreduce_span_ms"] > 0 and tr["exposed_ms"] >= 0, tr
    assert tr["wire_format"] == wire and tr["bytes_exchanged_per_step"] == step.exchange_bytes_per_step, tr
    # the exchanged gradient itself must be bit-identical on both ranks (bf16 wire: both unpack the same summed bf16 values)
    gb = [torch.empty_like(g2) for _ in range(world)]
    dist.all_gather(gb, g2)
    assert torch.equal(gb[0], gb[1]), "ranks hold different gradients after the exchange"

    # every rank must hold the same parameters afterwards
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "ranks diverged"

    if rank == 0:
        ref = model()
        one = sq_train.FusedTrainStep(ref, lr=1e-3, world_size=1)
        one.step(x, y)
        r1 = ref._gflat.clone()
        one.step(x[0:4], y[0:4])
        torch.cuda.synchronize()
        r2 = ref._gflat.clone()

        def rel(a, b):
            return float((a - b).abs().max() / b.abs().max())
        e1, e2 = rel(g1, r1), rel(g2, r2)
        dp = (flat - ref.flat.detach()).abs()
        # per tensor (every state_dict entry is a slice of the flat buffer)
        worst = 0.0
        for name, t in ref.state_dict().items():
            off = (t.data_ptr() - ref.flat.data_ptr()) // 4
            if 0 <= off < ref.flat.numel() and t.numel() and float(r1[off:off + t.numel()].abs().max()) > 0:
                worst = max(worst, rel(g1[off:off + t.numel()], r1[off:off + t.numel()]))
        print(f"ddp2 [{wire} wire, {step.exchange_bytes_per_step} B/step]: grad rel err step1 {e1:.2e} step2 {e2:.2e}, worst tensor {worst:.2e}; "
              f"param max diff {float(dp.max()):.2e} mean {float(dp.mean()):.2e}")
        if wire == "fp32":
            assert e1 < 1e-5 and e2 < 1e-5 and worst < 1e-4
            assert float(dp.max()) < 2.5e-3 and float(dp.mean()) < 1e-6      # lr = 1e-3: a sign flip of a ~zero gradient moves 2 lr at most
        else:               # two bf16 roundings (pack, sum): 2^-8 of a value at worst, relative to the tensor's maximum
            assert e1 < 1e-2 and e2 < 1e-2 and worst < 1e-2
            assert float(dp.max()) < 2.5e-3 and float(dp.mean()) < 2e-5
        open(sys.argv[1], "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
