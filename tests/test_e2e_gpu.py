"""End-to-end training step at VOC shape: 448x448 synthetic images through a
PyTorch-ROCm convolutional backbone (reduced width) and the HIP head; the
reference's call sequence of train.py:84-94, with and without GradScaler."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))

from oracle import ref_cpu
from yolo_amd import YOLOv1Head, synth

pytestmark = pytest.mark.gpu


def test_training_steps_reduce_the_loss(dev):
    from train_step import make_backbone

    torch.manual_seed(0)
    net = make_backbone(width=0.125).to(dev)
    head = YOLOv1Head(backbone=net)
    optim = torch.optim.AdamW(net.parameters(), lr=3e-4)
    image = torch.randn(4, 3, 448, 448, device=dev)
    _, gx, gc, om = (t.to(dev) for t in synth.make_inputs(4, seed=0))
    net.eval()                                           # dropout off: the loss must go down
    losses = []
    for _ in range(8):
        loss = head.get_loss(image, gx, gc, om)          # train.py:84
        optim.zero_grad()
        loss.backward()                                  # train.py:93
        optim.step()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]
    # the head's loss equals the oracle's on the backbone output
    with torch.no_grad():
        out = head(image)
        got = head.get_loss(image, gx, gc, om).item()
    want = ref_cpu.get_loss(out.cpu(), gx.cpu(), gc.cpu(), om.cpu()).item()
    assert abs(got - want) <= 1e-5 * abs(want)


def test_grad_scaler_sequence(dev):
    """scaler.scale(loss).backward(); scaler.step(optim); scaler.update() -- train.py:88-91."""
    from train_step import make_backbone

    torch.manual_seed(1)
    net = make_backbone(width=0.125).to(dev).eval()
    head = YOLOv1Head(backbone=net)
    optim = torch.optim.SGD(net.parameters(), lr=1e-3)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    image = torch.randn(2, 3, 448, 448, device=dev)
    _, gx, gc, om = (t.to(dev) for t in synth.make_inputs(2, seed=3))
    # reference gradient of the last layer's bias without scaling
    head.get_loss(image, gx, gc, om).backward()
    want = net[-1].bias.grad.clone()
    optim.zero_grad()
    loss = head.get_loss(image, gx, gc, om)
    scaler.scale(loss).backward()
    assert torch.allclose(net[-1].bias.grad, want * 1024.0, rtol=1e-6, atol=0)
    scaler.step(optim)
    scaler.update()
    assert torch.isfinite(net[-1].bias).all()


def test_half_precision_backbone_output(dev):
    """A backbone run under autocast hands over fp16: get_loss computes in fp32 and
    the gradient comes back in the backbone's dtype."""
    from yolo_amd import YOLOv1Head, synth
    head = YOLOv1Head()
    pred, gx, gc, om = (x.to(dev) for x in synth.make_inputs(8, seed=12))
    half = pred.half().requires_grad_(True)
    loss = head.get_loss(half, gx, gc, om)
    loss.backward()
    want = head.get_loss(half.detach().float(), gx, gc, om)
    assert loss.dtype == torch.float32 and abs(loss.item() - want.item()) <= 1e-6 * abs(want.item())
    assert half.grad is not None and half.grad.dtype == torch.float16 and torch.isfinite(half.grad).all()


def test_validation_under_inference_mode(dev):
    """train.py:100-110: validate() calls get_loss under torch.inference_mode()."""
    from yolo_amd import YOLOv1Head, synth
    head = YOLOv1Head()
    pred, gx, gc, om = (x.to(dev) for x in synth.make_inputs(8, seed=13))
    want = head.get_loss(pred, gx, gc, om).item()
    with torch.inference_mode():
        p2 = pred.clone()
        loss = head.get_loss(p2, gx, gc, om)
        dets, count = head.postprocess(p2, 0.5, 0.5)
    assert loss.item() == want and not loss.requires_grad
    assert int(count.sum()) > 0 and dets.shape == (8, 98, 6)


def test_evaluation_pass_matches_the_oracle(dev):
    """examples/eval_step.py: head_step (loss + detections) -> pooled rows -> get_map, all on the device, against
    the CPU oracle's post-processing and the reference's metric on the same rows."""
    import importlib.util
    import os

    import numpy as np

    from oracle import ref_cpu
    from yolo_amd import synth

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "eval_step.py")
    spec = importlib.util.spec_from_file_location("eval_step", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    head = YOLOv1Head()
    out = synth.make_pred(24, seed=4)
    ltrb, label, cnt = synth.make_gt_boxes(24, seed=5)
    m, pred, gt, loss = mod.evaluate(head, out.to(dev), ltrb.to(dev), label.to(dev), cnt.to(dev), 0.3, 0.5, 2.0)
    # the validation loss of the same launch (train.py:100-110) against the oracle on the
    # oracle's own target encoding
    gx, gc, om = ref_cpu.encode_targets(ltrb, label, cnt)
    want_loss = ref_cpu.get_loss(out, gx, gc, om)
    assert abs(loss.item() - want_loss.item()) <= 1e-5 * abs(want_loss.item())
    wd, wc, _ = ref_cpu.postprocess(out, 0.3, 0.5, mode=0)
    want_pred = np.concatenate([np.asarray(wd)[b, : int(wc[b])] for b in range(24)], axis=0)
    # the candidates' boxes and scores are decoded on the GPU (exp within a few ulp of the
    # CPU's), so the pooled rows are compared to tolerance and the metric on the GPU's own rows
    assert pred.shape == want_pred.shape
    np.testing.assert_allclose(pred.cpu().numpy(), want_pred, rtol=1e-5, atol=2e-4)
    assert m.item() == ref_cpu.get_map(pred.cpu(), gt.cpu(), 20, 2.0)
    assert m.item() > 0
