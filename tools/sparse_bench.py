import sys, os
sys.path.insert(0, "/root/repo")
import torch
from yolo_amd import YOLOv1Head, synth, MODE_REFERENCE, MODE_DENSE
dev = torch.device("cuda:0")
def b2b(fn, iters=100):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
for S, BX, C, B, mode in ((7, 2, 20, 4096, MODE_REFERENCE), (14, 2, 20, 1024, MODE_REFERENCE), (14, 5, 80, 1024, MODE_DENSE), (13, 5, 20, 1024, MODE_REFERENCE)):
    head = YOLOv1Head(32 * S, S, BX, C)
    pred = synth.make_inputs(B, S, BX, C, 32 * S)[0].to(dev)
    for thr in (0.001, 0.5, 0.9, 0.99, 0.999):
        t = b2b(lambda: head.postprocess(pred, thr, 0.5, mode))
        _, c = head.postprocess(pred, thr, 0.5, mode)
        print(f"S={S} B={BX} C={C} batch {B} mode {mode} conf>={thr}: {t:6.1f} us, kept/img {c.float().mean().item():.1f}")
