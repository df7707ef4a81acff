"""Times the split-mode 28x28 / 14x14 chain launch (chain_x3w.hip) in isolation against the two gemm_x3.hip launches it replaces,
under its ablation switches (sq_dbg_set key 1: 1 no global stores, 2 no identity reads, 4 no first product, 8 no second product).

    python tools/chainw_probe.py [n_patches] [dbg list, comma separated]"""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import _lib

lib = _lib.lib()
vp = ctypes.c_void_p
lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
lib.sq_dbg_chain_x3w.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong] + [vp] * 16 + [ctypes.c_int, vp]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dbgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 8, 12, 15]
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
P_ = _lib.ptr


def planes(*shape, scale=0.5):
    return (torch.randn(2, *shape, device=dev) * scale).to(torch.float16)


def timeit(fn, iters=6):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for C, N2, hw in ((256, 256, 14 * 14), (128, 128, 28 * 28), (128, 256, 28 * 28)):
    P = n * hw
    N1 = 4 * C
    t2, res, y, t1 = planes(P, C), planes(P, N1), planes(P, N1), planes(P, N2)
    wall = planes(N1 * C + N2 * N1, scale=0.05)
    w3, w1 = wall[:, :N1 * C].view(2, N1, C), wall[:, N1 * C:].view(2, N2, N1)
    w3c, w1c = w3.contiguous(), w1.contiguous()
    fp = torch.rand(4 * N1, device=dev) + 0.5
    b3, cs3, b1, cs1 = fp[:N1], fp[N1:2 * N1], fp[2 * N1:2 * N1 + N2], fp[3 * N1:3 * N1 + N2]

    def fused():
        rc = lib.sq_dbg_chain_x3w(1, C, N2, P, P_(t2[0]), P_(t2[1]), P_(res[0]), P_(res[1]), P_(y[0]), P_(y[1]), P_(t1[0]), P_(t1[1]),
                                  P_(w3[0]), P_(w3[1]), P_(w1[0]), P_(w1[1]), P_(b3), P_(cs3), P_(b1), P_(cs1), 0, st)
        assert rc == 0, rc

    def expand():
        _lib.check(lib.sq_linear_x3(1, P_(t2[0]), P_(t2[1]), C, P_(w3c[0]), P_(w3c[1]), C, P_(b3), P_(cs3), P_(res[0]), P_(res[1]), N1, 2,
                                    P_(y[0]), P_(y[1]), None, N1, P, N1, C, None, st))

    def reduce_():
        _lib.check(lib.sq_linear_x3(1, P_(y[0]), P_(y[1]), N1, P_(w1c[0]), P_(w1c[1]), N1, P_(b1), P_(cs1), None, None, N2, 2,
                                    P_(t1[0]), P_(t1[1]), None, N2, P, N2, N1, None, st))

    te, tr = timeit(expand), timeit(reduce_)
    flop = 2.0 * P * (N1 * C + N1 * N2)
    byts = P * 4.0 * (C + N1 + N1 + N2)
    print(f"C={C} N2={N2} P={P}: unfused expand {te:.1f} us + reduce {tr:.1f} us = {te + tr:.1f} us", flush=True)
    for d in dbgs:
        lib.sq_dbg_set(1, d)
        tf = timeit(fused)
        lib.sq_dbg_set(1, 0)
        print(f"   fused dbg{d}: {tf:8.1f} us   {flop / tf * 1e-6:7.1f} TF algorithmic ({3 * flop / tf * 1e-6:7.1f} TF of MFMA work)   {byts / tf * 1e-6:6.2f} TB/s algorithmic", flush=True)
