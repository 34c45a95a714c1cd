"""GPU box: latency of a tiny kernel + 32-byte device-to-host copy + wait, once per 0.5 s for 40 s (run right after another GPU process)."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(8, device=dev)
h = torch.zeros(8).pin_memory()
big = torch.empty(1 << 28, device=dev)  # 1 GB: something to allocate
t_start = time.perf_counter()
while time.perf_counter() - t_start < 40:
    lat = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x.add_(1.0)
        h.copy_(x, non_blocking=True)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e6)
    lat.sort()
    print("t=%5.1f s  read-back round trip: median %.0f us, max %.0f us" % (time.perf_counter() - t_start, lat[10], lat[-1]), flush=True)
    time.sleep(0.5)
