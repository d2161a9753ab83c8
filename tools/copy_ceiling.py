"""Calibration: duration of a plain device copy moving the same bytes as the
loss kernel (67.6 MB read+written) and as decode+NMS (33.7 MB)."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, iters=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
for total_mb in (33.7, 52.0, 67.6, 135.0, 540.0, 2000.0, 8000.0):
    n = int(total_mb * 1e6 / 2 / 4)
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    us = timeit(lambda: y.copy_(x))
    print(f"copy {total_mb:6.1f} MB total traffic: {us:7.2f} us  -> {total_mb / us:7.2f} TB/s")
    us = timeit(lambda: y.zero_())
    print(f"fill {total_mb/2:6.1f} MB written      : {us:7.2f} us  -> {total_mb / 2 / us:7.2f} TB/s")
