"""Worker of tests/test_gpu_resnet.py::test_halo_staged_3x3_matches_implicit_gemm: the bf16 embedder on 300 patches in a fresh
process (SQ_CONV_HALO is read once per process); saves the features to argv[1]."""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd import synth
from sequoia_pub_amd.resnet import resnet50
torch.manual_seed(5)
rn = resnet50(pretrained=False, compute_dtype="bf16").to("cuda:0").eval()
for m in rn.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
p = torch.from_numpy(synth.patches_u8(3, 300, 224)).cuda()
f = rn.extract_patches_u8(p, sub_batch=300)
torch.cuda.synchronize()
torch.save(f.cpu(), sys.argv[1])
print("features", f.shape, float(f.abs().mean()), bool(torch.isfinite(f).all()))
