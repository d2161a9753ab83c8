"""Measured error of the HIP loss kernel against the fp32 and fp64 CPU oracles at batch 4096."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu
from yolo_amd import YOLOv1Head, synth
dev = torch.device("cuda:0")
head = YOLOv1Head()
pred, gx, gc, om = synth.make_inputs(4096, seed=0)
terms, grad = head.loss_fwd_bwd(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev))
terms, grad = terms.cpu(), grad.cpu()
for name, dt in (("fp32 oracle", torch.float32), ("fp64 oracle", torch.float64)):
    ref = ref_cpu.loss_and_grad(pred, gx, gc, om, dtype=dt)
    rel = [abs(terms[i].item() - ref[k].item()) / abs(ref[k].item()) for i, k in enumerate(("coord", "conf", "cls", "total", "loss"))]
    g = ref["grad"].double()
    err = (grad.double() - g).abs()
    print(f"{name}: term rel err coord/conf/cls/total/loss = " + " ".join(f"{r:.2e}" for r in rel))
    print(f"{name}: grad max abs err {err.max().item():.3e} / max|grad| {g.abs().max().item():.3e} = {err.max().item() / g.abs().max().item():.2e}")
    nz = g.abs() > 1e-3 * g.abs().max()
    print(f"{name}: elementwise rel err over |grad| > 1e-3 max: max {(err[nz] / g.abs()[nz]).max().item():.2e}  median {(err[nz] / g.abs()[nz]).median().item():.2e}")
f32 = ref_cpu.loss_and_grad(pred, gx, gc, om)
f64 = ref_cpu.loss_and_grad(pred, gx, gc, om, dtype=torch.float64)
e = (f32["grad"].double() - f64["grad"]).abs().max().item() / f64["grad"].abs().max().item()
print(f"for scale: fp32 oracle vs fp64 oracle grad err {e:.2e}, loss rel {abs(f32['loss'].item() - f64['loss'].item()) / f64['loss'].item():.2e}")
