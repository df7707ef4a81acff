"""Where does the PCIe-inclusive pipeline lose time?  python tools/upload_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sequoia_pub_amd import synth
from sequoia_pub_amd.pipeline import SlidePipeline
from sequoia_pub_amd.resnet import resnet50
from sequoia_pub_amd.vis import ViS

dev = torch.device("cuda:0")
torch.manual_seed(99)
rn = resnet50(pretrained=False, compute_dtype="bf16").to(dev).eval()
vis = ViS(num_outputs=20820, input_dim=2048, depth=6, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64,
          device="cuda:0", compute_dtype="bf16").to(dev).eval()
pipe = SlidePipeline(rn, vis)
host = [torch.from_numpy(synth.patches_u8(i, 1000, 224)).pin_memory() for i in range(int(os.environ.get("NSL", "4")))]
resident = [h.to(dev) for h in host]
staging = [torch.empty_like(h, device=dev) for h in host]
cs = torch.cuda.Stream(device=dev)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def upload(wait_main=True):
    main = torch.cuda.current_stream(dev)
    if wait_main:
        cs.wait_stream(main)
    evs = []
    with torch.cuda.stream(cs):
        for h, d in zip(host, staging):
            d.copy_(h, non_blocking=True)
            e = torch.cuda.Event(); e.record(cs); evs.append(e)
    return evs


print("resident                      %.1f ms" % timed(lambda: pipe(resident)))
print("upload only                   %.1f ms" % timed(lambda: upload()))
print("upload + resident (no dep)    %.1f ms" % timed(lambda: (upload(), pipe(resident))))
print("upload -> staged (dependent)  %.1f ms" % timed(lambda: pipe(list(zip(staging, upload())))))
print("embed only, resident          %.1f ms" % timed(lambda: [pipe.embed(r) for r in resident]))
print("embed only, dependent upload  %.1f ms" % timed(lambda: [(torch.cuda.current_stream().wait_event(e), pipe.embed(d)) for d, e in zip(staging, upload())]))
