"""Same-box check that two builds of the library give the same bits: ResNet-50 f16x3 features of seeded synthetic patches.
    SQ_HIP_LIB=<build A> python tools/ab_features.py /tmp/a.pt;  python tools/ab_features.py /tmp/b.pt /tmp/a.pt"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd  # noqa
from sequoia_pub_amd import synth
from sequoia_pub_amd.resnet import resnet50

torch.manual_seed(5)
m = resnet50(compute_dtype="f16x3").to("cuda:0").eval()
for mod in m.modules():                        # BN statistics away from the identity so that scales / biases matter
    if isinstance(mod, torch.nn.BatchNorm2d):
        with torch.no_grad():
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5); mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.1)
outs = []
for n, size in ((24, 224), (5, 256)):
    p = torch.from_numpy(synth.patches_u8(7, n_patches=n, size=size)).cuda()
    outs.append(m.extract_patches_u8(p).cpu())
torch.save(outs, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for a, b in zip(outs, ref):
        print("bit-equal:", bool(torch.equal(a, b)), "finite:", bool(torch.isfinite(a).all()), "max abs diff", float((a - b).abs().max()), "max", float(a.abs().max()))
