"""Package power and clock while ONE split-fp16 kernel class of the headline step runs in a loop (rocm-smi sampled from the host while a few
seconds of launches are queued): where do the joules of a slide go?   python tools/power_by_class.py"""
import ctypes, os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import _lib

lib = _lib.lib()


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Package Power \(W\): ([0-9.]+)", out)
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else float("nan"), int(c.group(1)) if c else 0)


def run(name, M, N, K, conv=None, res=True, seconds=7.0, zeros=False):
    g = torch.Generator(device="cuda").manual_seed(1)
    if conv:
        n, H, Cin = conv
        A = torch.relu(torch.randn(2, n * H * H, Cin, device="cuda", generator=g)).half()
        A[1] *= 2.0 ** -11
        geom, lda = (ctypes.c_int * 9)(n, H, H, Cin, H, H, 3, 1, 1), 0
    else:
        A = torch.relu(torch.randn(2, M, K, device="cuda", generator=g)).half()     # post-ReLU activations: half of them zero, lo plane 2^-11 of hi
        A[1] *= 2.0 ** -11
        geom, lda = None, K
    W = torch.randn(2, N, K, device="cuda", generator=g).half()
    C = torch.empty(2, M, N, device="cuda", dtype=torch.float16)
    R = torch.randn(2, M, N, device="cuda", generator=g).half() if res else None
    if zeros:                                # the same launch on all-zero activations and weights: what is left is not multiplier switching
        A.zero_(); W.zero_()
        if res: R.zero_()
    bias = torch.randn(N, device="cuda")
    fn = lambda: _lib.check(lib.sq_linear_x3(1, _lib.ptr(A[0]), _lib.ptr(A[1]), lda, _lib.ptr(W[0]), _lib.ptr(W[1]), K, _lib.ptr(bias), None,
                                             _lib.ptr(R[0]) if res else None, _lib.ptr(R[1]) if res else None, N, 2,
                                             _lib.ptr(C[0]), _lib.ptr(C[1]), None, N, M, N, K, geom, _lib.stream_ptr()))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    n_iter = max(10, int(seconds * 1e3 / a.elapsed_time(b)))
    samples, stop = [], threading.Event()
    def sampler():                           # the enqueue loop below blocks on the queue depth: sample from a second thread
        t0 = time.time()
        while not stop.is_set():
            v = smi()
            if time.time() - t0 > seconds * 0.45:      # rocm-smi's figure is a running average: keep the second half of the run
                samples.append(v)
    th = threading.Thread(target=sampler); th.start()
    a.record()
    for _ in range(n_iter):
        fn()
    b.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    samples = samples[:-1] or samples
    us = a.elapsed_time(b) / n_iter * 1e3
    pw = sum(s[0] for s in samples) / len(samples)
    ck = sum(s[1] for s in samples) / len(samples)
    time.sleep(2.0)
    print(f"{name:34s} {us:8.1f} us  {pw:7.0f} W  sclk {ck:5.0f} MHz  {pw * us * 1e-6:7.3f} J per launch   {3 * 2.0 * M * N * K / us / 1e6:7.0f} TF of fp16 MFMA work", flush=True)


def run_torch(name, fn, work, unit, seconds=6.0):
    """the same sampling around a torch op (context figures: a plain HBM copy, the vendor library's fp16 / bf16 GEMM)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    n_iter = max(10, int(seconds * 1e3 / a.elapsed_time(b)))
    samples, stop = [], threading.Event()
    def sampler():
        t0 = time.time()
        while not stop.is_set():
            v = smi()
            if time.time() - t0 > seconds * 0.45:
                samples.append(v)
    th = threading.Thread(target=sampler); th.start()
    a.record()
    for _ in range(n_iter):
        fn()
    b.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    samples = samples[:-1] or samples
    us = a.elapsed_time(b) / n_iter * 1e3
    pw = sum(s[0] for s in samples) / len(samples); ck = sum(s[1] for s in samples) / len(samples)
    time.sleep(2.0)
    print(f"{name:34s} {us:8.1f} us  {pw:7.0f} W  sclk {ck:5.0f} MHz  {pw * us * 1e-6:7.3f} J per call     {work / us / 1e6:7.2f} {unit}", flush=True)


if __name__ == "__main__":
    print("idle:", smi())
    if len(sys.argv) > 1 and sys.argv[1] == "context":
        x = torch.randn(1 << 30, device="cuda"); y = torch.empty_like(x)                     # 4 GiB each
        run_torch("HBM copy 4 GiB -> 4 GiB (torch)", lambda: y.copy_(x), 2.0 * x.numel() * 4, "TB/s (read + write)")
        del x, y
        for dt, nm in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
            A = torch.randn(8192, 8192, device="cuda").to(dt); B = torch.randn(8192, 8192, device="cuda").to(dt)
            run_torch(f"hipBLASLt {nm} 8192^3 (torch.matmul)", lambda: torch.matmul(A, B.T), 2.0 * 8192 ** 3, "PFLOP/s x 1e-3" if False else "TFLOP/s")
            Az = torch.zeros_like(A)
            run_torch(f"  the same, A all zeros", lambda: torch.matmul(Az, B.T), 2.0 * 8192 ** 3, "TFLOP/s")
        sys.exit(0)
    run("3x3 14x14 (M196000 N256 K2304)", 196000, 256, 2304, conv=(1000, 14, 256), res=False)
    run("3x3 28x28 (M784000 N128 K1152)", 784000, 128, 1152, conv=(1000, 28, 128), res=False)
    run("reduce 14x14 (M196000 N256 K1024)", 196000, 256, 1024, res=False)
    run("reduce 28x28 (M784000 N128 K512)", 784000, 128, 512, res=False)
    run("expand 14x14 (M196000 N1024 K256)", 196000, 1024, 256, res=True)
    run("expand 28x28 (M784000 N512 K128)", 784000, 512, 128, res=True)
    run("3x3 14x14, all operands zero", 196000, 256, 2304, conv=(1000, 14, 256), res=False, zeros=True)
    run("reduce 14x14, all operands zero", 196000, 256, 1024, res=False, zeros=True)
    run("expand 28x28, all operands zero", 784000, 512, 128, res=True, zeros=True)
