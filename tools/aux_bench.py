"""Timing of the rows next to the hot path: target encoder, pairwise IoU, mAP."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolo_amd import YOLOv1Head, synth, box_ops
dev = torch.device("cuda:0")
def b2b(fn, iters=100):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
head = YOLOv1Head()
lt, lb, ct = (t.to(dev) for t in synth.make_gt_boxes(4096, seed=1))
lb32, ct32 = lb.to(torch.int32), ct.to(torch.int32)
t = b2b(lambda: head.encode_targets(lt, lb32, ct32))
print(f"encode_targets batch 4096: {t:.1f} us  ({4096 * 4753 / t / 1e3:.0f} GB/s written)")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g8_map.npz"))
pr, gt = torch.from_numpy(g["pred_a"]).float().to(dev), torch.from_numpy(g["gt_a"]).float().to(dev)
t = b2b(lambda: box_ops.get_map(pr, gt, 10, 0.5))
print(f"get_map P=1024 G=32 C=10: {t:.1f} us per call")
a = torch.rand(4096, 4, device=dev) * 448; b = torch.rand(4096, 4, device=dev) * 448
t = b2b(lambda: box_ops.get_iou(a, b))
print(f"get_iou 4096 x 4096: {t:.1f} us  ({4096 * 4096 * 4 / t / 1e3:.0f} GB/s written)")
