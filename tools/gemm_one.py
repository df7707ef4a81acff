"""Run one sq_linear shape a few times (for rocprofv3 --pmc passes)."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd import _lib
M, N, K = (int(x) for x in sys.argv[1:4])
lib = _lib.lib()
A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    _lib.check(lib.sq_linear(1, _lib.ptr(A), K, _lib.ptr(W), K, None, None, 0, 0, 0, _lib.ptr(C), 1, N, M, N, K, None, 0, _lib.stream_ptr()))
torch.cuda.synchronize()
