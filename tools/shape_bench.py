"""Loss and decode+NMS timings for shapes other than the benchmark one (they run
the run-time-shape kernels): back-to-back HIP-event microseconds per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, MODE_REFERENCE
dev = torch.device("cuda:0")
def b2b(fn, iters=100):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
for S, BX, C, B in ((7, 2, 20, 4096), (7, 2, 24, 4096), (7, 3, 20, 4096), (7, 2, 80, 4096), (14, 2, 20, 1024), (14, 5, 80, 1024), (13, 5, 20, 1024)):
    head = YOLOv1Head(32 * S, S, BX, C)
    pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(B, S, BX, C, 32 * S))
    D = 5 * BX + C
    lb = B * (2 * 4 * D * S * S + S * S * (4 * (4 + C) + 1))
    nb_ = B * (4 * D * S * S + 24 * S * S * BX + 4)
    tl = b2b(lambda: head.loss_fwd_bwd(pred, gx, gc, om))
    tn = b2b(lambda: head.postprocess(pred, 0.5, 0.5, MODE_REFERENCE))
    print(f"S={S:2d} B={BX} C={C:2d} batch {B}: loss {tl:7.1f} us {lb / tl / 1e3:7.0f} GB/s | decode+NMS {tn:7.1f} us {nb_ / tn / 1e3:7.0f} GB/s")
