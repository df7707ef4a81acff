import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sequoia_pub_amd
from sequoia_pub_amd import synth
from sequoia_pub_amd.kmeans import kmeans_fit_batch
X = torch.stack([torch.from_numpy(synth.features_gmm(0, 1000, 2048))]).cuda()
for _ in range(3): r = kmeans_fit_batch(X)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): r = kmeans_fit_batch(X)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"lone slide: {dt*1e3:.3f} ms, n_iter {r['n_iter'].tolist()}")
