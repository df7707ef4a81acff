"""Where do the ~23 s of a fresh ResNet50's first extract_patches_u8 go?  (bench.py's accuracy leg pays them once per weight set.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd import _lib, synth
from sequoia_pub_amd.resnet import resnet50, pack_weights, split_planes
from oracle import resnet_oracle as ro

def T(msg, t0):
    torch.cuda.synchronize(); print(f"{msg:50s} {time.perf_counter() - t0:7.2f} s", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "bind":
    from sequoia_pub_amd.cli.common import bind_to_gpu_numa_node
    print(bind_to_gpu_numa_node(0))
t0 = time.perf_counter(); h = torch.empty(1, dtype=torch.int32, pin_memory=True); T("first pinned allocation", t0)
for rep in range(2):
    print("--- instance", rep)
    t0 = time.perf_counter(); sd = ro.init_resnet50_state_dict(seed=99 + rep, perturb_bn=True); T("init state dict (CPU)", t0)
    t0 = time.perf_counter(); rn = resnet50(pretrained=False, compute_dtype="f16x3"); full = rn.state_dict(); full.update(sd); rn.load_state_dict(full); T("build + load_state_dict", t0)
    t0 = time.perf_counter(); rn = rn.to("cuda:0").eval(); T(".to(cuda)", t0)
    t0 = time.perf_counter(); w, b = pack_weights(rn.state_dict()); T("pack_weights (BN fold, fp64, from device tensors)", t0)
    t0 = time.perf_counter(); w2, b2 = split_planes(w, b, _lib.SQ_F16X3); T("split_planes", t0)
    t0 = time.perf_counter(); rn._pack(); T("rn._pack() (all of the above + upload)", t0)
    p = torch.from_numpy(synth.patches_u8(3, 1000, 224)).cuda()
    t0 = time.perf_counter(); f = rn.extract_patches_u8(p, sub_batch=500); T("first extract_patches_u8 (workspaces, kernels)", t0)
    t0 = time.perf_counter(); f = rn.extract_patches_u8(p, sub_batch=500); T("second extract_patches_u8", t0)
