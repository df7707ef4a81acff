"""Times the split-mode 56x56 tail / chain launch (chain_x3.hip) in isolation under its ablation switches
(sq_dbg_set key 1: 1 no stores, 2 no identity reads, 4 no 3x3, 8 no second product)."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import _lib

lib = _lib.lib()
lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
lib.sq_dbg_chain_x3.argtypes = [ctypes.c_int] * 4 + [ctypes.c_longlong, ctypes.c_int] + [ctypes.c_void_p] * 5
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
P = n * 56 * 56
dev = "cuda:0"
act = (torch.randn(2 * P * (64 + 256 + 256 + 128), device=dev) * 0.5).to(torch.float16).view(torch.int16)
wts = (torch.randn(2 * 102400, device=dev) * 0.05).to(torch.float16).view(torch.int16)
fp = torch.rand(2048, device=dev) + 0.5
frag = torch.zeros_like(wts)
st = torch.cuda.current_stream().cuda_stream


def run(n2, ds, tail, dbg, iters=6):
    lib.sq_dbg_set(1, dbg)
    for _ in range(2):
        rc = lib.sq_dbg_chain_x3(1, n2, ds, tail, P, 56, act.data_ptr(), wts.data_ptr(), fp.data_ptr(), frag.data_ptr(), st)
        assert rc == 0, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.sq_dbg_chain_x3(1, n2, ds, tail, P, 56, act.data_ptr(), wts.data_ptr(), fp.data_ptr(), frag.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    lib.sq_dbg_set(1, 0)
    return e0.elapsed_time(e1) / iters * 1e3


forms = [(64, 0, 1), (64, 1, 1), (128, 0, 1), (64, 0, 0)]
dbgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 8, 12, 15]
print("P", P, "form = (n2, ds, tail); us per launch")
for n2, ds, tail in forms:
    print((n2, ds, tail), "  ".join(f"dbg{d}={run(n2, ds, tail, d):7.1f}" for d in dbgs), flush=True)
