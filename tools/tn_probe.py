"""Weight-gradient (TN) GEMM timing vs K-slice count: python tools/tn_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequoia_pub_amd import _lib


def run(T, NO, NI, splits, bias=True):
    lib = _lib.lib()
    dY = torch.randn(T, NO, device="cuda").bfloat16()
    X = torch.randn(T, NI, device="cuda").bfloat16()
    dW = torch.empty(NO, NI, device="cuda")
    db = torch.empty(NO, device="cuda") if bias else None
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for sp in splits:
        lib.sq_dbg_set(4, sp)
        def call():
            _lib.check(lib.sq_linear_weight_grad(_lib.SQ_BF16, _lib.ptr(dY), NO, _lib.ptr(X), NI, _lib.ptr(dW), NI, _lib.ptr(db), NO, NI, T,
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"T={T} {NO}x{NI} split {sp:2d}: {us:7.1f} us  {2.0 * T * NO * NI / us / 1e6:7.1f} TF (kernel + reduce)")
    lib.sq_dbg_set(4, 0)


def run_group(T, NO, NI, members, ring, bias=True):
    """the four weight gradients of a ViS layer in one launch (sq_linear_weight_grad_group), ring form on / off"""
    import ctypes
    lib = _lib.lib()
    dY = [torch.randn(T, NO, device="cuda").bfloat16() for _ in range(members)]
    X = [torch.randn(T, NI, device="cuda").bfloat16() for _ in range(members)]
    dW = [torch.empty(NO, NI, device="cuda") for _ in range(members)]
    db = [torch.empty(NO, device="cuda") for _ in range(members)]
    arr = lambda ts: (ctypes.c_void_p * members)(*[t.data_ptr() for t in ts])
    lib.sq_dbg_set(15, ring)
    def call():
        _lib.check(lib.sq_linear_weight_grad_group(_lib.SQ_BF16, members, arr(dY), arr(X), arr(dW), arr(db) if bias else None, NO, NI, NI, NO, NI, T,
                                                   _lib.stream_ptr()))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    lib.sq_dbg_set(15, -1)
    print(f"group of {members}: T={T} {NO}x{NI} ring {ring} bias {int(bias)}: {us:7.1f} us  {2.0 * T * NO * NI * members / us / 1e6:7.1f} TF")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":       # short run for rocprofv3 --pmc passes
        run_group(6400, 1024, 1024, 4, 1, False)
        run_group(6400, 1024, 1024, 4, 0, False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "abl":       # ring form under its ablation switches (results not meaningful)
        for dbg in (0, 2, 4, 6):
            _lib.lib().sq_dbg_set(1, dbg)
            print(f"dbg {dbg:2d} (2 no requests, 4 no MFMA):", end=" ")
            run_group(6400, 1024, 1024, 4, 1, False)
        for dbg in (0, 2, 4, 6):
            _lib.lib().sq_dbg_set(1, dbg)
            print(f"dbg {dbg:2d} single gradient, 64 blocks:", end=" ")
            _lib.lib().sq_dbg_set(4, 1)
            run_group(6400, 1024, 1024, 1, 1, False)
            _lib.lib().sq_dbg_set(4, 0)
        _lib.lib().sq_dbg_set(1, 0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "group":
        for ring in (0, 1):
            for bias in (True, False):
                run_group(6400, 1024, 1024, 4, ring, bias)
        for ring in (0, 1):
            run_group(6400, 1024, 1024, 1, ring)
            run_group(25600, 1024, 1024, 4, ring)
            run_group(6400, 2048, 2048, 1, ring)
        for ring in (0, 1):
            _lib.lib().sq_dbg_set(15, ring)
            run(6400, 1024, 1024, (0, 1, 2, 4, 8))
        _lib.lib().sq_dbg_set(15, -1)
        sys.exit(0)
    run(6400, 1024, 1024, (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14))
    run(6400, 1024, 1024, (0, 4, 7), bias=False)
    # vendor reference for the same product (hipBLASLt through torch): dW = dY^T . X
    dY = torch.randn(6400, 1024, device="cuda").bfloat16(); X = torch.randn(6400, 1024, device="cuda").bfloat16()
    for _ in range(5): torch.matmul(dY.T, X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): torch.matmul(dY.T, X)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"torch.matmul(dY.T, X) bf16: {us:7.1f} us  {2.0 * 6400 * 1024 * 1024 / us / 1e6:7.1f} TF")
