import sys, os, time, torch
sys.path.insert(0, "/root/repo")
import sequoia_pub_amd
from sequoia_pub_amd import synth
from sequoia_pub_amd.kmeans import kmeans_fit_batch
for S, D in ((1, 2048), (8, 2048), (8, 1024)):
    X = torch.stack([torch.from_numpy(synth.features_gmm(i, 1000, D)) for i in range(S)]).cuda()
    for _ in range(2): r = kmeans_fit_batch(X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): r = kmeans_fit_batch(X)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"k-Means(100) on {S} slide(s) of 1000 x {D}: {dt*1e3:.2f} ms per call = {dt*1e3/S:.2f} ms/slide, n_iter {r['n_iter'].tolist()}")
