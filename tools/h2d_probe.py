import torch, time
x = torch.empty(150_528_000, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device="cuda")
for _ in range(2): d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"pinned H2D: {x.numel()/dt/1e9:.1f} GB/s ({dt*1e3:.1f} ms per 150.5 MB slide)")
