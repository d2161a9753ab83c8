"""yh_nms on candidates handed over in memory (batch 4096), back-to-back HIP-event timing.

    [YOLO_AMD_LIB=build/other.so] python tools/nms_only_bench.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head
dev = torch.device("cuda:0")
B = 4096
out = []
for n, ncls, ct in ((49, 20, 0.5), (49, 20, 0.0), (98, 20, 0.5), (98, 20, 0.0), (128, 2, 0.0), (64, 100, 0.0)):
    g = torch.Generator().manual_seed(n)
    c = torch.rand(B, n, 2, generator=g) * 448
    wh = torch.rand(B, n, 2, generator=g) * 200 + 4
    ltrb = torch.cat([(c - wh / 2).clamp(min=0), (c + wh / 2).clamp(max=448)], dim=-1).contiguous().to(dev)
    score = torch.rand(B, n, generator=g).to(dev)
    cls = torch.randint(0, ncls, (B, n), generator=g, dtype=torch.int32).to(dev)
    present = torch.ones(B, n, dtype=torch.bool).to(dev)
    fn = lambda: YOLOv1Head.nms(ltrb, score, cls, present, ct, 0.5)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    k = [e for e in prof.key_averages() if "nms_kernel" in e.key]
    us = sum(e.device_time_total for e in k) / 50
    out.append(f"n={n} classes={ncls} conf={ct}: {us:.2f} us (kept/image {fn()[1].float().mean().item():.1f})")
print(os.environ.get("YOLO_AMD_LIB", "in-tree"), *out, sep="\n  ")
