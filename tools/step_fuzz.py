"""yh_head_step (one kernel for loss fwd+bwd and decode+NMS at S=7, Bx=2, C=20 in reference
mode) on random batches, object densities, thresholds, row counts and logit scales: every
output compared bit for bit with yh_loss_fwd_bwd + yh_decode_nms_rows on the same inputs, and
every 20th configuration with the CPU oracle as well.  A development aid for changes to
step_kernels.hpp / loss_kernels.hpp / nms_kernels.hpp.

    python tools/step_fuzz.py [configs] [seed]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import ref_cpu
from yolo_amd import MODE_REFERENCE, YOLOv1Head, synth

dev = torch.device("cuda:0")
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
head = YOLOv1Head()
t0 = time.time()
kept = objects = checked = 0
for it_no in range(n_cfg):
    batch = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 33, 64, 129, 1000]))
    seed = int(rng.integers(0, 1 << 30))
    g = torch.Generator().manual_seed(seed)
    pred = synth.make_pred(batch, seed=seed) * float(rng.choice([0.1, 1.0, 1.0, 3.0, 20.0]))
    if rng.random() < 0.3:       # a crowded class
        pred[:, 10 + int(rng.integers(0, 20))] += float(rng.choice([2.0, 30.0]))
    density = float(rng.choice([0.0, 0.05, 0.1, 0.5, 0.8, 1.0]))
    om = torch.rand(batch, 49, 1, 1, generator=g) < density
    gx = (torch.rand(batch, 49, 1, 4, generator=g) * 0.98 + 0.01) * om
    gc = torch.softmax(torch.randn(batch, 49, 1, 20, generator=g) * 3, dim=-1) * om
    conf = float(rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])); iou = float(rng.choice([0.0, 0.3, 0.5, 0.9]))
    rows = [None, 49, 50, 98, 120][int(rng.integers(0, 5))]
    gb = None if rng.random() < 0.5 else int(batch * rng.integers(1, 9))
    d = [t.to(dev) for t in (pred, gx, gc, om)]
    terms, grad = head.loss_fwd_bwd(*d, global_batch=gb)
    dets, count, kidx = head.postprocess(d[0], conf, iou, MODE_REFERENCE, return_idx=True, rows=rows)
    got = head.head_step(*d, conf, iou, MODE_REFERENCE, global_batch=gb, rows=rows, return_idx=True)
    for name, a, b in zip(("terms", "grad", "dets", "count", "kidx"), got, (terms, grad, dets, count, kidx)):
        if not torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32)):
            print(f"MISMATCH {name}: config {it_no} batch {batch} seed {seed} density {density} conf {conf} iou {iou} rows {rows}")
            sys.exit(1)
    kept += int(count.sum()); objects += int(om.sum())
    if it_no % 20 == 0 and batch <= 129:
        ref = ref_cpu.loss_and_grad(pred, gx, gc, om, global_batch=gb)
        if torch.isfinite(ref["grad"]).all():      # the literal graph yields NaN where the kernels return the masked 0
            assert abs(terms[4].item() - ref["loss"].item()) <= 1e-5 * abs(ref["loss"].item()) + 1e-12, (it_no, terms, ref["loss"])
            scale = ref["grad"].abs().max().item()
            assert (grad.cpu() - ref["grad"]).abs().max().item() <= 1e-5 * scale + 1e-12, it_no
            checked += 1
print(f"{n_cfg} configurations equal ({time.time() - t0:.0f} s; {objects} object cells, {kept} rows kept, "
      f"{checked} configurations also against the oracle)")
