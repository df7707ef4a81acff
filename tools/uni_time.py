import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd.uni import create_model
from sequoia_pub_amd import synth
m = create_model(compute_dtype="bf16").to("cuda:0").eval()
p = torch.from_numpy(synth.patches_u8(0, 1000, 224)).cuda()
for sb in (64, 128, 256):
    m.extract_patches_u8(p[:sb], sub_batch=sb); torch.cuda.synchronize()
    t0 = time.perf_counter(); f = m.extract_patches_u8(p, sub_batch=sb); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"UNI ViT-L/16 bf16, 1000 patches, sub-batch {sb}: {dt*1e3:.1f} ms = {1000/dt:.0f} patches/s = {122.5e9*1000/dt/1e12:.0f} TFLOP/s (61.3 GMAC/patch)")
