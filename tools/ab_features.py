"""Bit-identity check between two builds of the library (same-box A/B): python tools/ab_features.py save <file> [S] / cmp <a> <b>.
`save` runs 96 structured patches through the split-fp16 ResNet-50 with the library SQ_HIP_LIB names (default: the in-tree one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    print("identical" if torch.equal(a, b) else f"DIFFERENT: max abs {float((a - b).abs().max()):.3e}")
    sys.exit(0 if torch.equal(a, b) else 1)

from oracle import resnet_oracle as ro          # weight recipe only
from sequoia_pub_amd import synth
from sequoia_pub_amd.resnet import resnet50

S = int(sys.argv[3]) if len(sys.argv) > 3 else 224
sd = ro.init_resnet50_state_dict(seed=3, perturb_bn=True)
m = resnet50(compute_dtype="f16x3")
m.load_state_dict(sd, strict=False)
m.to("cuda:0").eval()
patches = torch.from_numpy(synth.structured_patches_u8(0, 96, S, seed=5)).cuda()
with torch.no_grad():
    f = m.extract_patches_u8(patches, sub_batch=96)
torch.save(f.cpu(), sys.argv[2])
print("features", tuple(f.shape), float(f.abs().max()))
