"""Why does a single-round launch of the 256-wide GEMM kernels cost ~59 us in tools/gemm_probe.py when rocprofv3 sees a 24.5 us kernel?
Times N back-to-back sq_linear calls (wall clock around a synchronize, and events) with and without the split-K workspace."""
import ctypes, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd import _lib
lib = _lib.lib(); lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
M, N, K = 6400, 1024, 1024
A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16(); C = torch.empty(M, N, device="cuda")
WS = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
for tile, bn in ((22, -1), (88, 128), (88, 256), (55, -1)):
    for ws in (None, WS):
        lib.sq_dbg_set(0, tile); lib.sq_dbg_set(13, bn)
        call = lambda: _lib.check(lib.sq_linear(_lib.SQ_BF16, _lib.ptr(A), K, _lib.ptr(W), K, None, None, 0, 0, 0, _lib.ptr(C), 0, N, M, N, K,
                                                _lib.ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0, _lib.stream_ptr()))
        for _ in range(10): call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): call()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): call()
        b.record(); torch.cuda.synchronize()
        print(f"tile {tile} bn {bn} workspace {'yes' if ws is not None else 'no '}: host enqueue {t_host / 200 * 1e6:6.1f} us/call, wall {t_all / 200 * 1e6:6.1f} us/call, events {a.elapsed_time(b) / 200 * 1e3:6.1f} us/call")
