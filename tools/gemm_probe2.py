"""GPU experiment: memory-bound conv-like GEMMs (bf16 in/out, bf16 residual, ReLU) per tile config."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd import _lib
lib = _lib.lib()
lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]

def time_fn(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

def probe(M, N, K, res, tiles=(22, 21, 12, 11), dbgs=(0,)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16()
    R = torch.randn(M, N, device="cuda").bfloat16() if res else None
    bias = torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    byt = (M * K + N * K + M * N * (2 if res else 1)) * 2
    print(f"M={M} N={N} K={K} res={res}: {byt/1e6:.0f} MB", flush=True)
    for tile in tiles:
        for dbg in dbgs:
            lib.sq_dbg_set(0, tile); lib.sq_dbg_set(1, dbg)
            fn = lambda: _lib.check(lib.sq_linear(1, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(bias), _lib.ptr(R), N, 1, 2, _lib.ptr(C), 1, N, M, N, K, None, 0, _lib.stream_ptr()))
            t = time_fn(fn)
            print(f"   tile {tile} dbg {dbg}: {t:8.1f} us  {byt / t / 1e6:7.2f} TB/s  {2.0*M*N*K/t/1e6:7.1f} TF", flush=True)
    lib.sq_dbg_set(0, 0); lib.sq_dbg_set(1, 0)

if __name__ == "__main__":
    probe(627200, 256, 64, True, dbgs=(0, 1, 2))
    probe(627200, 256, 64, False, dbgs=(0, 1))
    probe(156800, 512, 128, True)
    probe(39200, 1024, 256, True)
    probe(627200, 64, 256, False)
    # plain device copy for reference
    x = torch.empty(627200 * 256, device="cuda", dtype=torch.bfloat16); y = torch.empty_like(x)
    t = time_fn(lambda: y.copy_(x)); print(f"copy 321MB->321MB: {t:.1f} us {x.numel()*4/t/1e6:.2f} TB/s")
