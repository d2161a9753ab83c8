"""Seeded sweep over head shapes, batch sizes, thresholds and modes: every
configuration runs the loss (against the oracle, tolerance 1e-5 as in
test_loss_gpu.py), the fused decode+NMS and the stand-alone NMS (against the
oracle on the same candidates, bit for bit), and the target encoder (bit for
bit).  The shapes are drawn so that every launch variant is hit: one wavefront
per image with one or two candidates per lane, one workgroup per image with
1, 2 or 4 candidates per thread, class counts that are and are not multiples
of four, more cells than threads in a workgroup.  YH_SWEEP_CONFIGS sets the
number of configurations, YH_SWEEP_WIDE=1 widens the ranges (grids up to 30,
16 predictors, 300 classes, 2,048 candidates)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import MODE_DENSE, MODE_REFERENCE, YOLOv1Head, synth

pytestmark = pytest.mark.gpu


def configs():
    rng = np.random.default_rng(20240607)
    out = []
    wide = os.environ.get("YH_SWEEP_WIDE") == "1"     # larger grids, up to 16 predictors, more classes
    while len(out) < int(os.environ.get("YH_SWEEP_CONFIGS", "64")):
        S = int(rng.integers(1, 31 if wide else 15))
        Bx = int(rng.integers(1, 17 if wide else 6))
        C = int(rng.choice([1, 2, 3, 4, 7, 20, 21, 33, 64, 65, 80, 128, 129, 300] if wide
                           else [1, 2, 3, 4, 7, 20, 21, 33, 80, 100]))
        batch = int(rng.choice([1, 2, 3, 5, 9, 17]))
        mode = int(rng.integers(0, 2))
        ct = float(rng.choice([0.001, 0.1, 0.3, 0.5, 0.7]))
        it = float(rng.choice([0.2, 0.45, 0.5, 0.7]))
        out.append((S, Bx, C, batch, mode, ct, it, len(out)))
    return out


@pytest.mark.parametrize("S,Bx,C,batch,mode,ct,it,seed", configs())
def test_random_configuration(dev, S, Bx, C, batch, mode, ct, it, seed):
    img = 32 * S
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    hp = dict(n_cells=S, n_bboxes=Bx, n_classes=C, img_size=img)
    pred, gx, gc, om = synth.make_inputs(batch, S, Bx, C, img, seed=1000 + seed)
    # loss
    terms, grad = head.loss_fwd_bwd(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev))
    ref = ref_cpu.loss_and_grad(pred, gx, gc, om, **hp)
    for i, name in enumerate(("coord", "conf", "cls", "total", "loss")):
        want = ref[name].item()
        assert abs(terms[i].item() - want) <= 1e-5 * max(abs(want), 1e-30), name
    scale = ref["grad"].abs().max().item()
    assert (grad.cpu() - ref["grad"]).abs().max().item() <= 1e-5 * scale
    # decode + NMS, fused and stand-alone, on the candidates the GPU decoded
    got = head.postprocess(pred.to(dev), ct, it, mode, return_idx=True)
    ltrb, score, cls, present = head.decode_candidates(pred.to(dev), mode)
    want = ref_cpu.nms_batch(ltrb.cpu().numpy(), score.cpu().numpy(), cls.cpu().numpy(),
                             present.cpu().numpy(), ct, it)
    for g_, w in zip(got, want):
        np.testing.assert_array_equal(g_.cpu().numpy().view(np.uint32), np.asarray(w).view(np.uint32))
    # the same tensor in channels-last memory format (cell-major) is read in place: same
    # bits from the run-time-shape kernels, the same rows from decode + NMS
    pcl = pred.to(dev).contiguous(memory_format=torch.channels_last)
    terms_cl, grad_cl = head.loss_fwd_bwd(pcl, gx.to(dev), gc.to(dev), om.to(dev))
    if (S, Bx, C) != (7, 2, 20):
        assert torch.equal(terms_cl, terms) and torch.equal(grad_cl.contiguous(), grad)
    if S * S * Bx <= 2048:
        for g_, w in zip(head.postprocess(pcl, ct, it, mode, return_idx=True), got):
            assert torch.equal(g_, w)
    got2 = YOLOv1Head.nms(ltrb, score, cls, present, ct, it, return_idx=True)
    for g_, w in zip(got2, want):
        np.testing.assert_array_equal(g_.cpu().numpy().view(np.uint32), np.asarray(w).view(np.uint32))
    # target encoder
    gl, lb, cnt = synth.make_gt_boxes(batch, C, img, seed=2000 + seed)
    ex, ec, em = head.encode_targets(gl.to(dev), lb.to(dev), cnt.to(dev))
    ox, oc, oo = ref_cpu.encode_targets(gl, lb, cnt, S, C, img)
    assert torch.equal(ex.cpu(), ox) and torch.equal(ec.cpu(), oc) and torch.equal(em.cpu(), oo)
