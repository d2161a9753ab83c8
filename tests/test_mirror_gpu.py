"""The reference's remaining head methods under their own names (model.py:150-222,
224-294) through the C ABI, against the reference fixtures and the oracle.
Element-wise box transforms: bit-exact (same fp32 operations in the same order).
Sigmoid-based values: 1e-6 absolute (GPU exponential)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import YOLOv1Head, synth

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_box_transforms_against_reference_fixture(dev, golden):
    g = golden("g2_decode.npz")
    head = YOLOv1Head()
    cm = t(g["norm_cellmajor"]).view(1, 49, 2, 4).to(dev)
    den = head.denormalize_xywh(cm)
    assert torch.equal(den.cpu().view(98, 4), t(g["denorm"]))
    assert torch.equal(head.denormalize_xywh(cm[0]).cpu().view(98, 4), t(g["denorm"]))   # one image
    assert torch.equal(head.xywh_to_ltrb(den).cpu().view(98, 4), t(g["ltrb"]))
    assert torch.equal(head.xywh_to_ltrb(t(g["extra_xywh"]).to(dev)).cpu(), t(g["extra_ltrb"]))
    pred = t(g["pred_img"])[None].to(dev)
    norm = head.get_norm_xywh(pred).cpu()[0]
    ref = t(g["norm_boxmajor"])
    assert (norm[:, 0] - ref[:49]).abs().max().item() <= 1e-6
    assert (norm[:, 1] - ref[49:]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("S,Bx,C,img,batch", [(7, 2, 20, 448, 257), (14, 5, 80, 448, 3), (3, 1, 4, 96, 9)])
def test_transforms_and_chain_against_oracle(dev, S, Bx, C, img, batch):
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_pred(batch, S, Bx, C, seed=S)
    norm = ref_cpu.get_norm_xywh(pred, Bx)
    den = ref_cpu.denormalize_xywh(norm, S, img)
    assert torch.equal(head.denormalize_xywh(norm.to(dev)).cpu(), den)
    assert torch.equal(head.xywh_to_ltrb(den.to(dev)).cpu(), ref_cpu.xywh_to_ltrb(den, img))
    got = head.get_norm_xywh(pred.to(dev))
    assert (got.cpu() - norm).abs().max().item() <= 1e-6
    # the chain equals get_ltrb bit for bit (same device arithmetic either way)
    chain = head.xywh_to_ltrb(head.denormalize_xywh(got))
    assert torch.equal(chain, head.get_ltrb(pred.to(dev)))


def test_object_mask_against_reference_fixture(dev, golden):
    g = golden("g4_objmask.npz")
    head = YOLOv1Head()
    gt, cnt, mask = t(g["gt_ltrb"]), t(g["count"]), t(g["mask"])
    for i in range(gt.shape[0]):
        got = head.get_object_mask(gt[i, :int(cnt[i])].contiguous().to(dev))
        assert torch.equal(got.cpu(), mask[i])
    assert not head.get_object_mask(torch.zeros(0, 4, device=dev)).any()


@pytest.mark.parametrize("S,Bx,C,img,batch", [(7, 2, 20, 448, 512), (14, 5, 80, 448, 4), (7, 3, 20, 448, 33)])
def test_responsibility_mask_and_terms(dev, S, Bx, C, img, batch):
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred, gx, gc, om = synth.make_inputs(batch, S, Bx, C, img, seed=31 + S)
    hp = dict(n_cells=S, n_bboxes=Bx, n_classes=C, img_size=img)
    coord, conf, cls, resp = ref_cpu.loss_terms(pred, gx, gc, om, **hp)
    got = head._get_responsibility_mask(pred.to(dev), gx.to(dev), om.to(dev)).cpu()
    assert got.shape == resp.shape and got.dtype == torch.bool
    # exactly one predictor per object cell, none elsewhere
    assert torch.equal(got.sum(dim=2) > 0, om[:, :, 0])
    assert int(got.sum(dim=2).max()) <= 1
    # the oracle's choice, bit for bit: the kernel forms the IoU with the oracle's
    # operation order and an IEEE division (eval.py:61-65)
    assert torch.equal(got, resp)
    tc, tf, tk = head.get_loss_terms(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev))
    for g_, w in ((tc, coord), (tf, conf), (tk, cls)):
        assert abs(g_.item() - w.item()) <= 1e-5 * abs(w.item())
    assert head.get_coordinate_loss(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev)).item() == tc.item()
    assert head.get_confidence_loss(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev)).item() == tf.item()
    assert head.get_classification_loss(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev)).item() == tk.item()
