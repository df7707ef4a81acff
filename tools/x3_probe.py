"""GPU experiment: time sq_linear_x3 (split-fp16 product, gemm_x3.hip) per block shape and ablation switch.
Not part of the product or the tests.   python tools/x3_probe.py [shapes]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import _lib

lib = _lib.lib()
lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]


def time_fn(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def probe(M, N, K, conv=None, res=True, shapes=(0, 1 << 30), dbgs=(0, 1, 2, 4, 8, 6, 12)):
    g = torch.Generator(device="cuda").manual_seed(1)
    if conv:
        n, H, Cin = conv
        A = (torch.randn(2, n * H * H, Cin, device="cuda", generator=g)).half()
        geom = (ctypes.c_int * 9)(n, H, H, Cin, H, H, 3, 1, 1)
        lda = 0
    else:
        A = torch.randn(2, M, K, device="cuda", generator=g).half()
        geom, lda = None, K
    W = torch.randn(2, N, K, device="cuda", generator=g).half()
    C = torch.empty(2, M, N, device="cuda", dtype=torch.float16)
    R = torch.randn(2, M, N, device="cuda", generator=g).half() if res else None
    bias = torch.randn(N, device="cuda")
    flops = 2.0 * M * N * K
    for sh in shapes:
        for dbg in dbgs:
            lib.sq_dbg_set(7, sh)
            lib.sq_dbg_set(1, dbg)
            fn = lambda: _lib.check(lib.sq_linear_x3(1, _lib.ptr(A[0]), _lib.ptr(A[1]), lda, _lib.ptr(W[0]), _lib.ptr(W[1]), K, _lib.ptr(bias), None,
                                                     _lib.ptr(R[0]) if res else None, _lib.ptr(R[1]) if res else None, N, 2,
                                                     _lib.ptr(C[0]), _lib.ptr(C[1]), None, N, M, N, K, geom, _lib.stream_ptr()))
            t = time_fn(fn)
            print(f"M={M} N={N} K={K} {'conv3x3' if conv else 'gemm'} rows{'128' if sh else '256'} dbg {dbg:2d}: {t:8.1f} us  {flops / t / 1e6:7.1f} TF eff  {3 * flops / t / 1e6:7.1f} TF mfma", flush=True)
    lib.sq_dbg_set(7, -1)
    lib.sq_dbg_set(1, 0)


def probe_interleaved(M, N, K):
    """Hypothesis test: a K-tile's hi and lo 64-byte pieces of a row in ONE 128-byte line (dbg 16: K-tile stride doubled)."""
    A = torch.randn(M, 2 * K, device="cuda").half()
    W = torch.randn(N, 2 * K, device="cuda").half()
    C = torch.empty(2, M, N, device="cuda", dtype=torch.float16)
    bias = torch.randn(N, device="cuda")
    flops = 2.0 * M * N * K
    for dbg in (16,):
        lib.sq_dbg_set(1, dbg)
        fn = lambda: _lib.check(lib.sq_linear_x3(1, _lib.ptr(A), A.data_ptr() + 64, 2 * K, _lib.ptr(W), W.data_ptr() + 64, 2 * K, _lib.ptr(bias), None,
                                                 None, None, N, 2, _lib.ptr(C[0]), _lib.ptr(C[1]), None, N, M, N, K, None, _lib.stream_ptr()))
        t = time_fn(fn)
        print(f"M={M} N={N} K={K} gemm INTERLEAVED rows256 dbg {dbg:2d}: {t:8.1f} us  {flops / t / 1e6:7.1f} TF eff  {3 * flops / t / 1e6:7.1f} TF mfma", flush=True)
    lib.sq_dbg_set(7, -1)
    lib.sq_dbg_set(1, 0)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "add":
        # do the memory time and the matrix time of an HBM-bound expand add up?  (dbg 4 no MFMA, 1 no stores, 8 no fragment reads)
        for M, N, K in [(784000, 512, 128), (196000, 1024, 256)]:
            probe(M, N, K, res=True, shapes=(-1,), dbgs=(0, 0, 4, 12, 1, 5, 13))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "halo":
        # per-tile fixed cost of the halo-staged 3x3: the three stages' shapes with stores / loads switched off
        probe(784000, 128, 1152, conv=(1000, 28, 128), res=False, shapes=(-1,), dbgs=(0, 0, 1, 4, 5))
        probe(196000, 256, 2304, conv=(1000, 14, 256), res=False, shapes=(-1,), dbgs=(0, 0, 1, 4, 5))
        probe(49000, 512, 4608, conv=(1000, 7, 512), res=False, shapes=(-1,), dbgs=(0, 0, 1, 4, 5))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "res":
        # identity rows requested under the last K-tile (default) against requested in the epilogue (dbg 64), 128-row shape
        for M, N, K in [(784000, 512, 128), (196000, 1024, 256), (3136000, 256, 64), (392000, 512, 128), (98000, 1024, 256)]:
            probe(M, N, K, res=True, shapes=(-1,), dbgs=(0, 64, 0, 64))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "il":
        shapes = [(98000, 256, 2304), (98000, 256, 1024), (24500, 512, 4608)]
        if len(sys.argv) > 2:
            shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
        for M, N, K in shapes:
            probe(M, N, K, res=False, shapes=(-1,), dbgs=(0,))
            probe_interleaved(M, N, K)
        sys.exit(0)
    probe(98000, 256, 2304, conv=(500, 14, 256), res=False)
    probe(98000, 256, 2304, res=False, shapes=(0,))
    probe(98000, 256, 1024, res=False, shapes=(0,))
    probe(98000, 1024, 256, res=True)
    probe(24500, 512, 4608, res=False, shapes=(0,), dbgs=(0, 2, 4))
