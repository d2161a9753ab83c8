"""A cell-major head tensor -- the memory of a channels-last (B, D, S, S) tensor, for
backbones that end in a channels-last convolution instead of the reference's Linear +
view (model.py:133-135, 146-148) -- is read in place: yh_loss_fwd_bwd_nhwc and
yh_decode_nms_nhwc against the oracle and against the channel-major entry points."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import MODE_DENSE, MODE_REFERENCE, YOLOv1Head, _lib, synth

pytestmark = pytest.mark.gpu

SHAPES = [(7, 2, 20, 37), (14, 5, 80, 9), (5, 3, 7, 11), (13, 5, 20, 6), (1, 1, 1, 3), (16, 1, 3, 4)]


def channels_last(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("S,Bx,C,batch", SHAPES)
def test_loss_on_a_channels_last_tensor(dev, S, Bx, C, batch):
    img = 32 * S
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    hp = dict(n_cells=S, n_bboxes=Bx, n_classes=C, img_size=img)
    pred, gx, gc, om = synth.make_inputs(batch, S, Bx, C, img, seed=77 + S)
    d = [t.to(dev) for t in (pred, gx, gc, om)]
    out_cl = channels_last(d[0])
    terms, grad = head.loss_fwd_bwd(out_cl, *d[1:])
    ref = ref_cpu.loss_and_grad(pred, gx, gc, om, **hp)
    for i, name in enumerate(("coord", "conf", "cls", "total", "loss")):
        want = ref[name].item()
        assert abs(terms[i].item() - want) <= 1e-5 * max(abs(want), 1e-30), name
    scale = ref["grad"].abs().max().item()
    assert (grad.cpu() - ref["grad"]).abs().max().item() <= 1e-5 * scale
    # the gradient comes back in the tensor's own memory format
    assert grad.stride() == out_cl.stride()
    # against the channel-major call: the run-time-shape kernel does the same arithmetic per
    # cell in the same order, so every bit agrees (the tuned S=7,B=2,C=20 kernel adds its
    # loss rows in another order: there the terms agree to rounding)
    terms_cm, grad_cm = head.loss_fwd_bwd(*d)
    if (S, Bx, C) == (7, 2, 20):
        assert torch.allclose(terms, terms_cm, rtol=1e-6, atol=0)
        assert (grad - grad_cm).abs().max().item() <= 1e-6 * scale
    else:
        assert torch.equal(terms, terms_cm)
        assert torch.equal(grad.contiguous(), grad_cm)


def test_backward_reaches_a_channels_last_leaf(dev):
    head = YOLOv1Head()
    pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(8, seed=5))
    leaf = channels_last(pred).clone().requires_grad_(True)
    assert not leaf.is_contiguous() and leaf.is_contiguous(memory_format=torch.channels_last)
    loss = head.get_loss(leaf, gx, gc, om)
    (loss * 3.0).backward()
    want = ref_cpu.loss_and_grad(pred.cpu(), gx.cpu(), gc.cpu(), om.cpu())
    assert abs(loss.item() - want["loss"].item()) <= 1e-5 * want["loss"].item()
    scale = want["grad"].abs().max().item()
    assert (leaf.grad.cpu() - 3.0 * want["grad"]).abs().max().item() <= 3e-5 * scale


@pytest.mark.parametrize("S,Bx,C,batch", SHAPES)
def test_decode_nms_on_a_channels_last_tensor(dev, S, Bx, C, batch):
    img = 32 * S
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_inputs(batch, S, Bx, C, img, seed=99 + S)[0].to(dev)
    for mode, ct in ((MODE_REFERENCE, 0.5), (MODE_DENSE, 0.001), (MODE_DENSE, 0.6)):
        want = head.postprocess(pred, ct, 0.5, mode, return_idx=True)
        got = head.postprocess(channels_last(pred), ct, 0.5, mode, return_idx=True)
        for g_, w in zip(got, want):          # rows, counts and indices: the same bits
            np.testing.assert_array_equal(g_.cpu().numpy().view(np.uint32), w.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("S,Bx,C", [(14, 5, 80), (13, 5, 20), (9, 3, 130), (7, 2, 1000)])
def test_class_ties_and_nans_when_lanes_share_a_cell(dev, S, Bx, C):
    """A block too large to stage: 16 lanes take a cell's class logits together.  Equal
    maxima (the first index wins), -inf rows and NaNs (only one at class 0 counts) must
    come out as when one lane walks the cell (the channel-major kernel)."""
    img = 32 * S
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    g = torch.Generator().manual_seed(S * 1000 + C)
    batch = 6
    pred = synth.make_inputs(batch, S, Bx, C, img, seed=3)[0]
    cls = pred[:, 5 * Bx:]                                  # (batch, C, S, S), a view
    cls.copy_(torch.randint(-3, 4, cls.shape, generator=g).float())   # many equal maxima
    cls[1].fill_(float("-inf"))
    cls[2, 0] = float("nan")                                # NaN at class 0: class 0
    cls[3, C // 2] = float("nan")                           # elsewhere: passed over
    cls[3, min(17, C - 1), ::2] = float("inf")
    cls[4, :, 1::2, :] = 2.5                                # a whole cell equal
    cls[5, C - 1] = 7.0                                     # the last class, in the last lane's share
    pred = pred.to(dev)
    for mode, ct in ((MODE_REFERENCE, 0.3), (MODE_DENSE, 0.001)):
        want = head.postprocess(pred, ct, 0.5, mode, return_idx=True)
        got = head.postprocess(channels_last(pred), ct, 0.5, mode, return_idx=True)
        for g_, w in zip(got, want):
            np.testing.assert_array_equal(g_.cpu().numpy().view(np.uint32), w.cpu().numpy().view(np.uint32))
    # and the classes themselves against the oracle's argmax on one image
    dets, cnt, idx = head.postprocess(channels_last(pred), 0.0, 2.0, MODE_DENSE, return_idx=True)
    want_cls = ref_cpu_classes(pred[5].cpu(), Bx)
    got_cls = torch.empty(S * S, dtype=torch.long)
    k = int(cnt[5].item())
    got_cls[(idx[5, :k].cpu().long() // Bx)] = dets[5, :k, 5].cpu().long()
    assert torch.equal(got_cls, want_cls)


def ref_cpu_classes(pred_img, Bx):
    """argmax over the class channels per cell, row-major cells (model.py:389)."""
    return torch.argmax(pred_img[5 * Bx:], dim=0).reshape(-1)


def test_argument_errors_and_limits(dev):
    lib = _lib.load()
    f = 0x1000
    assert lib.yh_loss_fwd_bwd_nhwc(None, f, f, f, f, f, f, 1 << 20, 4, 7, 2, 20, 448, 5.0, 0.5, 0.25, None) == _lib.YH_ERR_NULL
    assert lib.yh_loss_fwd_bwd_nhwc(f, f, f, f, f, f, f, 1 << 20, 4, 7, 2, 20, 449, 5.0, 0.5, 0.25, None) == _lib.YH_ERR_SHAPE
    assert lib.yh_loss_fwd_bwd_nhwc(f, f, f, f, f, f, None, 0, 4, 7, 2, 20, 448, 5.0, 0.5, 0.25, None) == _lib.YH_ERR_WORKSPACE
    assert lib.yh_decode_nms_nhwc(f, f, f, None, 3, 0.5, 0.5, 4, 7, 2, 20, 448, None) == _lib.YH_ERR_ARG
    assert lib.yh_decode_nms_nhwc(f, f, f, None, 0, 0.5, 0.5, 4, 32, 3, 20, 448, None) == _lib.YH_ERR_UNSUPPORTED
    assert lib.yh_decode_nms_nhwc(f, f, f, None, 0, 0.5, 0.5, 0, 7, 2, 20, 448, None) == _lib.YH_OK
    # the Python mirror refuses what the cell-major entry points do not take
    head = YOLOv1Head()
    pred = synth.make_inputs(2, seed=1)[0].to(dev)
    with pytest.raises(ValueError):
        head.postprocess(channels_last(pred), 0.5, 0.5, bias=torch.zeros(1470, device=dev))
