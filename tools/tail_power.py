"""Are the 56x56 tails of the split-fp16 ResNet (chain_x3.hip) joule-bound or schedule-bound?  (round-6 review, item 2.)
Each tail class alone in a loop, ~6 s per arm, rocm-smi sampled from a second thread:
  arm A  the launch as the pipeline makes it (random post-ReLU operands)
  arm B  the same launch, the same bytes, ALL operands zero (no multiplier / operand-bus switching)
  arm C  stores disabled (sq_dbg_set(1, 1)), random operands
  arm D  zero operands and stores disabled
If B is faster than A at the same bytes, the class is limited by power (clock pulled down); if time and clock do not move, the
schedule is what is left.     python tools/tail_power.py [patches=1000]"""
import ctypes
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import _lib

lib = _lib.lib()
lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
lib.sq_dbg_chain_x3.argtypes = [ctypes.c_int] * 4 + [ctypes.c_longlong, ctypes.c_int] + [ctypes.c_void_p] * 5
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
P = n * 56 * 56
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Package Power \(W\): ([0-9.]+)", out)
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else float("nan"), int(c.group(1)) if c else 0)


def operands(zero):
    if zero:
        act = torch.zeros(2 * P * (64 + 256 + 256 + 128), device=dev, dtype=torch.int16)
        wts = torch.zeros(2 * 102400, device=dev, dtype=torch.int16)
        fp = torch.zeros(2048, device=dev)
    else:
        g = torch.Generator(device=dev).manual_seed(1)
        parts = []
        for c in (64, 256, 256, 128):                     # t1, identity, y, next t1: each a hi plane followed by its lo plane
            hi = torch.relu(torch.randn(P * c, device=dev, generator=g) * 0.5)                          # post-ReLU: half the values zero
            parts += [hi.to(torch.float16), (hi * 2.0 ** -11).to(torch.float16)]                         # lo plane: 2^-11 of hi
            del hi
        act = torch.cat(parts).view(torch.int16)
        del parts
        wts = (torch.randn(2 * 102400, device=dev, generator=g) * 0.05).to(torch.float16).view(torch.int16)
        fp = torch.rand(2048, device=dev, generator=g) + 0.5
    return act, wts, fp, torch.zeros_like(wts)


def arm(name, form, zero, dbg, seconds=6.0):
    n2, ds, tail = form
    act, wts, fp, frag = operands(zero)
    lib.sq_dbg_set(1, dbg)
    call = lambda: lib.sq_dbg_chain_x3(1, n2, ds, tail, P, 56, act.data_ptr(), wts.data_ptr(), fp.data_ptr(), frag.data_ptr(), st)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); call(); b.record(); torch.cuda.synchronize()
    n_iter = max(10, int(seconds * 1e3 / a.elapsed_time(b)))
    samples, stop = [], threading.Event()

    def sampler():
        t0 = time.time()
        while not stop.is_set():
            v = smi()
            if time.time() - t0 > seconds * 0.45:        # rocm-smi's power is a running average: keep the second half
                samples.append(v)
    th = threading.Thread(target=sampler)
    th.start()
    a.record()
    for _ in range(n_iter):
        call()
    b.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    lib.sq_dbg_set(1, 0)
    samples = samples[:-1] or samples
    us = a.elapsed_time(b) / n_iter * 1e3
    pw = sum(s[0] for s in samples) / max(len(samples), 1)
    ck = sum(s[1] for s in samples) / max(len(samples), 1)
    print(f"  {name:44s} {us:8.1f} us  {pw:6.0f} W  sclk {ck:5.0f} MHz  {pw * us * 1e-6:6.3f} J per launch  ({len(samples)} samples)", flush=True)
    del act, wts, fp, frag
    torch.cuda.empty_cache()
    time.sleep(2.0)
    return us, pw, ck


if __name__ == "__main__":
    print(f"idle: {smi()};  {n} patches per launch, P = {P} pixels")
    classes = [("tail_f16x3_c64_cn64   (n2 64, plain)", (64, 0, 1)), ("tail_f16x3_c64_cn64_ds (n2 64, downsample)", (64, 1, 1)),
               ("tail_f16x3_c64_cn128  (n2 128)", (128, 0, 1))]
    for cname, form in classes:
        print(cname)
        ra = arm("A random operands", form, False, 0)
        rb = arm("B all operands zero, same bytes", form, True, 0)
        rc = arm("C random operands, stores disabled", form, False, 1)
        rd = arm("D zero operands, stores disabled", form, True, 1)
        print(f"  -> zero operands: time x{rb[0] / ra[0]:.3f}, clock x{rb[2] / max(ra[2], 1):.3f}, power {ra[1]:.0f} -> {rb[1]:.0f} W;  "
              f"no stores: time x{rc[0] / ra[0]:.3f}", flush=True)
