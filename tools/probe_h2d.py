"""Developer probe: host -> device copy rate from pinned memory (torch, one stream / two streams / per-tensor sizes)."""
import time
import torch

dev = torch.device("cuda:0")
for mb in (12, 24, 96, 288):
    h = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32).pin_memory()
    d = torch.empty_like(h, device=dev)
    s = torch.cuda.Stream()
    for name, fn in (("copy_", lambda: d.copy_(h, non_blocking=True)), ("to()", lambda: h.to(dev, non_blocking=True))):
        with torch.cuda.stream(s):
            fn(); s.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            s.synchronize()
            dt = (time.perf_counter() - t0) / 10
        print(f"{mb:4d} MB {name:6s}: {mb / 1024 / dt:6.1f} GB/s ({dt * 1e3:.2f} ms)")
# two streams concurrently
h1 = torch.empty(96 * 1024 * 1024 // 4).pin_memory(); h2 = torch.empty(96 * 1024 * 1024 // 4).pin_memory()
d1 = torch.empty_like(h1, device=dev); d2 = torch.empty_like(h2, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
    with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"two streams x 96 MB: {2 * 96 / 1024 / dt:6.1f} GB/s")
# unpinned for comparison
hu = torch.empty(96 * 1024 * 1024 // 4)
torch.cuda.synchronize(); t0 = time.perf_counter(); d1.copy_(hu); torch.cuda.synchronize()
print(f"unpinned 96 MB: {96 / 1024 / (time.perf_counter() - t0):6.1f} GB/s")
import numpy as np
from smart_tree_amd.synthetic import sample_tree_cloud
t0 = time.perf_counter(); c = sample_tree_cloud(1_000_000, seed=5); print(f"sample_tree_cloud(1M): {time.perf_counter() - t0:.2f} s")
