"""Measured error of the HIP loss kernel against the fp32 and fp64 CPU oracles at
batch 4096: scalars, max-norm, and element by element per channel group, with the
fp32 oracle's own distance from fp64 beside every figure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu
from yolo_amd import YOLOv1Head, synth

dev = torch.device("cuda:0")
batch = int(os.environ.get("ACC_BATCH", "4096"))
head = YOLOv1Head()
pred, gx, gc, om = synth.make_inputs(batch, seed=0)
terms, grad = head.loss_fwd_bwd(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev))
terms, grad = terms.cpu(), grad.cpu().double()
f32 = ref_cpu.loss_and_grad(pred, gx, gc, om)
f64 = ref_cpu.loss_and_grad(pred, gx, gc, om, dtype=torch.float64)
names = ("coord", "conf", "cls", "total", "loss")
for who, val in (("hip", lambda i, k: terms[i].item()), ("fp32 oracle", lambda i, k: f32[k].item())):
    rel = [abs(val(i, k) - f64[k].item()) / abs(f64[k].item()) for i, k in enumerate(names)]
    print(f"{who:12s} vs fp64: term rel err coord/conf/cls/total/loss = " + " ".join(f"{r:.2e}" for r in rel))
g64, g32 = f64["grad"], f32["grad"].double()
obj = om[:, :, 0, 0].view(batch, 1, 7, 7).expand(batch, 30, 7, 7)
groups = {"xy": [0, 1, 5, 6], "wh": [2, 3, 7, 8], "conf": [4, 9], "cls": list(range(10, 30))}
print(f"{'group':10s} {'scale':>10s} | max-norm err hip / fp32 | elementwise rel err (entries > 1e-3 scale): hip max / fp32 max | "
      f"hip median / fp32 median | entries over max(2*|f32-f64|, 1e-5*|f64|)")
for name, chans in groups.items():
    for part, mask in (("obj", obj[:, chans]), ("noobj", ~obj[:, chans])):
        a, b, r = grad[:, chans][mask], g32[:, chans][mask], g64[:, chans][mask]
        if r.numel() == 0 or r.abs().max() == 0:
            continue
        scale = r.abs().max().item()
        eh, eo = (a - r).abs(), (b - r).abs()
        big = r.abs() > 1e-3 * scale
        over = (eh > torch.maximum(2 * eo, 1e-5 * r.abs())) & big
        print(f"{name + '/' + part:10s} {scale:10.3e} | {eh.max().item() / scale:.2e} / {eo.max().item() / scale:.2e} | "
              f"{(eh[big] / r.abs()[big]).max().item():.2e} / {(eo[big] / r.abs()[big]).max().item():.2e} | "
              f"{(eh[big] / r.abs()[big]).median().item():.2e} / {(eo[big] / r.abs()[big]).median().item():.2e} | "
              f"{int(over.sum())} of {int(big.sum())}")
