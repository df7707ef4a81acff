"""Launch target for rocprofv3 --pmc: the layer-3 3x3 split-fp16 convolution, halo-staged and implicit GEMM, and a long-K 1x1."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import _lib
lib = _lib.lib()
lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(1)
n, H, Cin, N = 500, 14, 256, 256
M, K = n * H * H, 9 * Cin
A = torch.randn(2, M, Cin, device="cuda", generator=g).half()
W = torch.randn(2, N, K, device="cuda", generator=g).half()
C = torch.empty(2, M, N, device="cuda", dtype=torch.float16)
bias = torch.randn(N, device="cuda")
geom = (ctypes.c_int * 9)(n, H, H, Cin, H, H, 3, 1, 1)
A2 = torch.randn(2, M, 1024, device="cuda", generator=g).half()
W2 = torch.randn(2, N, 1024, device="cuda", generator=g).half()
for halo in (-1, 0):
    lib.sq_dbg_set(8, halo)
    for _ in range(3):
        _lib.check(lib.sq_linear_x3(1, _lib.ptr(A[0]), _lib.ptr(A[1]), 0, _lib.ptr(W[0]), _lib.ptr(W[1]), K, _lib.ptr(bias), None, None, None, N, 2,
                                    _lib.ptr(C[0]), _lib.ptr(C[1]), None, N, M, N, K, geom, _lib.stream_ptr()))
for _ in range(3):
    _lib.check(lib.sq_linear_x3(1, _lib.ptr(A2[0]), _lib.ptr(A2[1]), 1024, _lib.ptr(W2[0]), _lib.ptr(W2[1]), 1024, _lib.ptr(bias), None, None, None, N, 2,
                                _lib.ptr(C[0]), _lib.ptr(C[1]), None, N, M, N, 1024, None, _lib.stream_ptr()))
torch.cuda.synchronize()
