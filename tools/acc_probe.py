import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from sequoia_pub_amd import synth, _lib
from oracle import resnet_oracle as ro, vis_oracle
from sequoia_pub_amd.pipeline import SlidePipeline
from sequoia_pub_amd.resnet import resnet50
from sequoia_pub_amd.vis import ViS
dev = torch.device("cuda:0")
def T(msg, t0):
    torch.cuda.synchronize(); print(f"{msg:60s} {time.perf_counter() - t0:7.2f} s", flush=True)
cfg = dict(bench.VIS_CFG, input_dim=2048)
t0 = time.perf_counter()
vis = ViS(**cfg, device=str(dev), compute_dtype="fp32"); vis.load_state_dict(vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=31), seed=32)); vis = vis.to(dev).eval(); T("ViS build", t0)
for rep, mk in enumerate((lambda: ro.init_resnet50_state_dict(seed=99, perturb_bn=True), lambda: ro.init_resnet50_state_dict(seed=7, perturb_bn=True))):
    sd = mk(); rn = resnet50(pretrained=False, compute_dtype="f16x3"); full = rn.state_dict(); full.update(sd); rn.load_state_dict(full); rn = rn.to(dev).eval()
    pipe = SlidePipeline(rn, vis, n_clusters=100, sub_batch=500)
    p = torch.from_numpy(synth.patches_u8(7, 1000, 224)).to(dev)
    for it in range(2):
        t0 = time.perf_counter(); f = pipe.embed(p); T(f"net {rep} call {it}: embed", t0)
        t0 = time.perf_counter(); cf, lab = pipe.cluster(f.unsqueeze(0)); T(f"net {rep} call {it}: cluster", t0)
        t0 = time.perf_counter(); pr = vis(cf); T(f"net {rep} call {it}: vis", t0)
        t0 = time.perf_counter(); out = pipe([p]); T(f"net {rep} call {it}: whole pipe([p])", t0)
print("--- after allocating 40 GB, freeing it and emptying the cache")
big = [torch.empty(8 << 30, dtype=torch.uint8, device=dev) for _ in range(5)]
torch.cuda.synchronize(); del big
t0 = time.perf_counter(); torch.cuda.empty_cache(); T("empty_cache (40 GB)", t0)
sd = ro.init_resnet50_state_dict(seed=5, perturb_bn=True); rn = resnet50(pretrained=False, compute_dtype="f16x3"); full = rn.state_dict(); full.update(sd); rn.load_state_dict(full); rn = rn.to(dev).eval()
pipe = SlidePipeline(rn, vis, n_clusters=100, sub_batch=500)
t0 = time.perf_counter(); f = pipe.embed(p); T("fresh net: embed", t0)
t0 = time.perf_counter(); x = torch.empty(8 << 30, dtype=torch.uint8, device=dev); T("torch.empty 8 GB", t0)
t0 = time.perf_counter(); x = None; torch.cuda.empty_cache(); T("free 8 GB", t0)
print("--- with the roofline profiler switched on and off (bench.py does that before the accuracy leg)")
_lib.prof_enable(True); f = pipe.embed(p); torch.cuda.synchronize(); r = _lib.prof_report(); _lib.prof_enable(False)
sd = ro.init_resnet50_state_dict(seed=6, perturb_bn=True); rn = resnet50(pretrained=False, compute_dtype="f16x3"); full = rn.state_dict(); full.update(sd); rn.load_state_dict(full); rn = rn.to(dev).eval()
pipe = SlidePipeline(rn, vis, n_clusters=100, sub_batch=500)
t0 = time.perf_counter(); out = pipe([p]); T("fresh net after profiling: pipe([p])", t0)
