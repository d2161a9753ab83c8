"""Timing of the NMS stress configuration (S=14, Bx=5, C=80, conf 0.001, dense
candidates: 980 boxes per image) and of the loss at that shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, MODE_DENSE, MODE_REFERENCE
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_BATCH", "1024"))
S, BX, C, IMG = 14, 5, 80, 448
head = YOLOv1Head(IMG, S, BX, C)
pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(B, S, BX, C, IMG))
def b2b(fn, iters=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
per_img_nms = 4 * 105 * 196 + 24 * 980 + 4
per_img_loss = 2 * 4 * 105 * 196 + 196 * (4 + 80) * 4 + 196
t = b2b(lambda: head.postprocess(pred, 0.001, 0.5, MODE_DENSE))
d, c = head.postprocess(pred, 0.001, 0.5, MODE_DENSE); torch.cuda.synchronize()
print(f"dense NMS   batch {B}: {t:8.1f} us  {B / t:7.2f} M img/s  {per_img_nms * B / t / 1e3:7.1f} GB/s  kept/img {c.float().mean().item():.1f}")
t = b2b(lambda: head.postprocess(pred, 0.5, 0.5, MODE_REFERENCE))
print(f"ref-mode NMS batch {B}: {t:8.1f} us  {B / t:7.2f} M img/s")
t = b2b(lambda: head.loss_fwd_bwd(pred, gx, gc, om))
print(f"loss        batch {B}: {t:8.1f} us  {B / t:7.2f} M img/s  {per_img_loss * B / t / 1e3:7.1f} GB/s")
# detections of real images crowd into few classes: one class wins a share of the cells
for share in (0.1, 0.3, 0.7):
    crowded = pred.clone()
    mask = torch.rand(B, S, S, device=dev, generator=torch.Generator(dev).manual_seed(1)) < share
    crowded[:, 5 * BX + 40][mask] += 30.0
    t = b2b(lambda: head.postprocess(crowded, 0.001, 0.5, MODE_DENSE))
    d, c = head.postprocess(crowded, 0.001, 0.5, MODE_DENSE); torch.cuda.synchronize()
    print(f"dense NMS, one class in {int(share * 100)} % of the cells: {t:8.1f} us  kept/img {c.float().mean().item():.1f}")
