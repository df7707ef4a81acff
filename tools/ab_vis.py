"""Same-box check that two builds of the library give the same bits on the ViS paths: bf16 / fp32 forward, two fused training steps.
    SQ_HIP_LIB=<build A> python tools/ab_vis.py /tmp/a.pt;  python tools/ab_vis.py /tmp/b.pt /tmp/a.pt"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import synth
from sequoia_pub_amd.vis import ViS
from sequoia_pub_amd.train import FusedTrainStep

cfg = dict(num_outputs=2000, input_dim=1024, depth=3, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64)
outs = {}
x = torch.from_numpy(synth.cluster_tokens(3, 48, 1024)).cuda()
y = torch.from_numpy(synth.rna_targets(4, 48, 2000)).cuda()
for mode in ("bf16", "fp32"):
    torch.manual_seed(11)
    m = ViS(**cfg, device="cuda:0", compute_dtype=mode).to("cuda:0")
    with torch.no_grad():
        outs[mode + "_fwd"] = m(x).float().cpu()
    st = FusedTrainStep(m, lr=1e-3)
    for _ in range(2):
        loss, pred, _ = st.step(x, y)
    outs[mode + "_params"] = m.flat.detach().float().cpu()
    outs[mode + "_pred"] = pred.float().cpu()
torch.save(outs, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for k in outs:
        print(k, "bit-equal:", bool(torch.equal(outs[k], ref[k])), "finite:", bool(torch.isfinite(outs[k]).all()), "max abs diff", float((outs[k] - ref[k]).abs().max()))
