"""The ViS training step's main-chain product (6400 x 1024 x 1024, bf16) per tile shape WITH the epilogues it has in the step (fp32 residual in,
fp32 result out) and operands rotated over 8 buffer sets (cold caches, as in the step): python tools/nt_small.py"""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd import _lib
lib = _lib.lib(); lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
M, N, K = 6400, 1024, 1024
NB = 8
A = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(NB)]
W = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(NB)]
R = [torch.randn(M, N, device="cuda") for _ in range(NB)]
C = [torch.empty(M, N, device="cuda") for _ in range(NB)]
bias = torch.randn(N, device="cuda")


def timeit(call, n=25):
    for i in range(NB): call(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(n):
        for i in range(NB): call(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * NB) * 1e3


for tile, bn in ((22, -1), (12, -1), (21, -1), (11, -1), (88, 128), (33, -1)):
    lib.sq_dbg_set(0, tile); lib.sq_dbg_set(13, bn)
    plain = lambda i: _lib.check(lib.sq_linear(_lib.SQ_BF16, _lib.ptr(A[i]), K, _lib.ptr(W[i]), K, None, None, 0, 0, 0, _lib.ptr(C[i]), 0, N, M, N, K, None, 0, _lib.stream_ptr()))
    resid = lambda i: _lib.check(lib.sq_linear(_lib.SQ_BF16, _lib.ptr(A[i]), K, _lib.ptr(W[i]), K, _lib.ptr(bias), _lib.ptr(R[i]), N, 0, 0, _lib.ptr(C[i]), 0, N, M, N, K, None, 0, _lib.stream_ptr()))
    try:
        print(f"tile {tile:2d} bn {bn:3d}: plain fp32 out {timeit(plain):6.1f} us   bias + fp32 residual, fp32 out {timeit(resid):6.1f} us")
    except Exception as e:
        print(f"tile {tile}: {e}")
lib.sq_dbg_set(0, 0); lib.sq_dbg_set(13, -1)
mm = lambda i: torch.matmul(A[i], W[i].T)
print(f"torch.matmul (hipBLASLt), bf16 out, nothing else: {timeit(mm):6.1f} us")
ad = lambda i: torch.addmm(R[i].bfloat16(), A[i], W[i].T)
