"""A/B of the dual launches' tile walk (gemm_x3.hip DUAL form; sq_dbg_set key 16: 0 = 4 x 16 strips per XCD, 8 = 8 x 8 squares):
per-class HIP-event times of one 1000-patch embed in the split-fp16 mode, features bit-compared.  Round-6 review item 3c: the A/B
that round 5 asserted instead of running.     python tools/dual_walk_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import _lib, synth
from sequoia_pub_amd.resnet import resnet50

lib = _lib.lib()
lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
os.environ["SQ_RESNET_STREAMS"] = "1"
torch.manual_seed(99)
rn = resnet50(pretrained=False, compute_dtype="f16x3").to("cuda:0").eval()
p = torch.from_numpy(synth.patches_u8(7, 1000, 224)).cuda()
feats = {}
for rep in range(2):
    for walk in (0, 8):
        lib.sq_dbg_set(16, walk)
        rn.extract_patches_u8(p)
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(3):
            f = rn.extract_patches_u8(p)
        recs = _lib.prof_report()
        _lib.prof_enable(False)
        feats[walk] = f.clone()
        tot = sum(r["total_ms"] for r in recs) / 3
        duals = {r["name"]: r["total_ms"] / r["count"] * 1e3 for r in recs if r["name"].startswith("dual_")}
        print(f"walk {walk}: all classes {tot:7.3f} ms per 1000 patches; " + "  ".join(f"{k} {v:7.1f} us" for k, v in sorted(duals.items())), flush=True)
lib.sq_dbg_set(16, 0)
print("features bit-identical between the walks:", bool(torch.equal(feats[0], feats[8])))
