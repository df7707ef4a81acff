"""GPU experiment: time sq_linear (MFMA NT GEMM engine) per tile config / ablation switch and
against torch.matmul (hipBLASLt) on the same operands.  Not part of the product or the tests."""
import ctypes
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd import _lib

lib = _lib.lib()
lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
WS = torch.empty(128 << 20, dtype=torch.uint8, device='cuda')        # split-K scratch (the launcher ignores it for forced special kernels)


def time_fn(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def probe(M, N, K, dtype, tiles=(22, 21, 12, 11), dbgs=(0, 1, 2, 4, 3, 6), out_bf16=False, scheds=(None,)):
    tdt = torch.bfloat16 if dtype == _lib.SQ_BF16 else torch.float32
    A = torch.randn(M, K, device="cuda").to(tdt)
    W = torch.randn(N, K, device="cuda").to(tdt)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    flops = 2.0 * M * N * K
    t = time_fn(lambda: torch.matmul(A, W.T))
    print(f"M={M} N={N} K={K} {tdt}: torch.matmul {t:8.1f} us {flops / t / 1e6:8.1f} TF", flush=True)
    for tile in tiles:
      for sched in (scheds if tile == 88 else (None,)):
        for dbg in dbgs:
            lib.sq_dbg_set(0, tile)
            lib.sq_dbg_set(1, dbg)
            lib.sq_dbg_set(10, -1 if sched is None else sched)
            fn = lambda: _lib.check(lib.sq_linear(dtype, _lib.ptr(A), K, _lib.ptr(W), K, None, None, 0, 0, 0, _lib.ptr(C), 1 if out_bf16 else 0, N, M, N, K, _lib.ptr(WS), WS.numel(), _lib.stream_ptr()))
            t = time_fn(fn)
            err = ""
            if dbg == 0 and M * N <= (1 << 28):
                ref = torch.matmul(A, W.T).float()
                err = f"  max rel err vs torch {float((C.float() - ref).abs().max() / ref.abs().max()):.2e}"
            print(f"   tile {tile}{'' if sched is None else ' sched %d' % sched} dbg {dbg}: {t:8.1f} us {flops / t / 1e6:8.1f} TF{err}", flush=True)
    lib.sq_dbg_set(0, 0)
    lib.sq_dbg_set(1, 0)
    lib.sq_dbg_set(10, -1)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        for M, N, K in [(8192, 8192, 8192), (102400, 1024, 1024), (98000, 1024, 256), (98000, 512, 1024), (24500, 2048, 512), (392000, 512, 128), (98000, 256, 1024), (392000, 128, 512)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(22,), dbgs=(0,))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "w4t":
        # gemm_w4.hip with K-tile-major operand addressing (dbg 32: weights, 96: weights and activations), timing only
        for M, N, K in [(8192, 8192, 8192), (102400, 1024, 1024), (50432, 4096, 1024), (50432, 1024, 4096), (50432, 3072, 1024), (102400, 2048, 2048)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(55,), dbgs=(0, 32, 96, 0, 32, 96))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "p8":
        # gemm_p8.hip (tile 88: 256 x 256 x 64, eight phases) against the engine's pick (0), gemm_w4.hip (55) and hipBLASLt
        for M, N, K in [(8192, 8192, 8192), (4096, 4096, 4096), (50432, 4096, 1024), (50432, 1024, 4096), (50432, 3072, 1024), (50432, 1024, 1024),
                        (102400, 2048, 2048), (102400, 1024, 1024), (6400, 1024, 1024), (24500, 2048, 1024)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(0, 55, 88, 88), dbgs=(0,), scheds=(0, 1))
        for M, N, K in [(50432, 4096, 1024), (50432, 3072, 1024)]:           # bf16 output (what the UNI blocks write)
            probe(M, N, K, _lib.SQ_BF16, tiles=(55, 88), dbgs=(0,), out_bf16=True, scheds=(0, 1))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "p8m":
        # mid-size products: where does the 256 x 256 kernel (tile 88) start to beat the engine's pick (tile 0 = the kernel it replaces
        # below its minimum of 176 tiles)?  tiles of 256 x 256 in brackets
        lib.sq_dbg_set(14, 0)          # gemm_p8.hip off (was the SQ_GEMM_P8=0 environment switch until round 6)
        for M, N, K in [(4096, 4096, 4096), (6400, 1024, 1024), (12800, 1024, 1024), (25600, 1024, 1024), (16384, 2048, 2048), (8192, 4096, 1024),
                        (24500, 2048, 1024), (24500, 512, 2048), (12800, 2048, 2048)]:
            print("tiles", ((M + 255) // 256) * (N // 256))
            probe(M, N, K, _lib.SQ_BF16, tiles=(0, 88, 0, 88), dbgs=(0,), scheds=(1,), out_bf16=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "p8s":
        # why is a single round of tiles slow?  fixed vs per-K-tile cost (K sweep) and the ablation switches on 6400 x 1024 x K
        for bn in (128, 256):
            lib.sq_dbg_set(13, bn)
            for K in (256, 1024, 4096):
                print("bn", bn)
                probe(6400, 1024, K, _lib.SQ_BF16, tiles=(88,), dbgs=(0, 1, 2, 4, 8, 14), scheds=(1,))
        lib.sq_dbg_set(13, -1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "p8n":
        # the 256 x 128 shape of gemm_p8.hip (forced width 128) on the ViS training step's products, against the engine's other kernels
        for M, N, K in [(6400, 1024, 1024), (6400, 1024, 1024), (12800, 1024, 1024), (3200, 1024, 1024), (6400, 2048, 1024)]:
            lib.sq_dbg_set(13, 128)
            probe(M, N, K, _lib.SQ_BF16, tiles=(22, 88, 22, 88), dbgs=(0,), scheds=(1,))
            probe(M, N, K, _lib.SQ_BF16, tiles=(88,), dbgs=(0,), scheds=(1,), out_bf16=True)
        lib.sq_dbg_set(13, -1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "p8p":
        # persistent form of gemm_p8.hip (sched 1) against one block per tile (sched 0)
        for out_bf16 in (False, True):
            for M, N, K in [(50432, 4096, 1024), (50432, 1024, 4096), (50432, 3072, 1024), (50432, 1024, 1024), (102400, 1024, 1024), (102400, 2048, 2048), (8192, 8192, 8192)]:
                probe(M, N, K, _lib.SQ_BF16, tiles=(88,), dbgs=(0,), scheds=(0, 1, 0, 1), out_bf16=out_bf16)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "p8g":
        # tile-walk group height of gemm_p8.hip (1 = row-major inside an XCD's run)
        for M, N, K in [(8192, 8192, 8192), (50432, 4096, 1024), (50432, 1024, 4096), (50432, 3072, 1024), (102400, 2048, 2048), (102400, 1024, 1024)]:
            for gm in (1, 2, 4, 8, 1, 4):
                lib.sq_dbg_set(11, gm)
                print("group_m", gm)
                probe(M, N, K, _lib.SQ_BF16, tiles=(88,), dbgs=(0,), scheds=(1,), out_bf16=True)
        lib.sq_dbg_set(11, -1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "p8a":
        # ablation of gemm_p8.hip: 1 no stores, 2 no LDS-DMA in the loop, 4 no MFMA, 8 no fragment reads
        for M, N, K in [(8192, 8192, 8192), (50432, 4096, 1024)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(88,), dbgs=(0, 1, 2, 4, 8, 6, 12, 10, 14, 0), scheds=(1,))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "p8e":
        # what of the epilogue is the HBM write burst: 1 no epilogue read-out / math / stores, 64 everything but the global stores,
        # 128 the stores of every tile onto tile (0, 0) (no HBM traffic); bf16 results
        for M, N, K in [(50432, 4096, 1024), (204800, 1024, 1024)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(88,), dbgs=(32, 1, 64, 128, 32, 1, 64, 128), scheds=(1,), out_bf16=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "w4":
        for M, N, K in [(8192, 8192, 8192), (4096, 4096, 4096), (102400, 1024, 1024), (50432, 4096, 1024), (50432, 1024, 4096), (24500, 512, 4608),
                        (24500, 2048, 1024), (98000, 1024, 512), (98000, 512, 1024), (6400, 1024, 1024)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(0, 55, 33), dbgs=(0,))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ring":
        for M, N, K in [(8192, 8192, 8192), (102400, 1024, 1024), (98000, 256, 2304), (98000, 256, 1024), (98000, 512, 1024),
                        (24500, 512, 4608), (24500, 2048, 1024), (392000, 128, 1152), (98000, 1024, 256), (6400, 1024, 1024)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(22, 33), dbgs=(0,))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "uni":
        for M, N, K in [(50432, 4096, 1024), (50432, 1024, 4096), (50432, 3072, 1024), (50432, 1024, 1024), (102400, 2048, 2048)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(22, 33), dbgs=(0,))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "l4":
        for M, N, K in [(24500, 512, 4608), (24500, 2048, 512), (24500, 512, 2048), (24500, 2048, 1024), (98000, 512, 1024), (6400, 1024, 1024)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(0, 22, 21, 12, 33), dbgs=(0,))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "resnet":
        for M, N, K in [(39200, 1024, 256), (9800, 2048, 512), (39200, 256, 1024), (156800, 512, 128), (156800, 128, 512),
                        (9800, 512, 2048), (627200, 64, 256), (627200, 256, 64), (39200, 512, 1024)]:
            probe(M, N, K, _lib.SQ_BF16, tiles=(0, 22, 21, 12, 11), dbgs=(0,))
        sys.exit(0)
    probe(6400, 1024, 1024, _lib.SQ_BF16, tiles=(22,), dbgs=(0, 1))
    probe(6400, 1024, 1024, _lib.SQ_F32, tiles=(22,), dbgs=(0,))
    probe(8192, 8192, 8192, _lib.SQ_BF16, tiles=(22,), dbgs=(0,))
    probe(39200, 256, 2304, _lib.SQ_BF16, tiles=(22, 21), dbgs=(0,))
    probe(9800, 512, 4608, _lib.SQ_BF16, tiles=(22,), dbgs=(0,))
    probe(156800, 128, 512, _lib.SQ_BF16, tiles=(0, 22), dbgs=(0,))
