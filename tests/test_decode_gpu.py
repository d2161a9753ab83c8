"""Grid decode helpers (the reference's inference-side methods) on the GPU
against fixtures captured from the reference itself (golden g2/g3) and against
the oracle.  Floating point: sigmoid/softmax values within 1e-6 absolute,
pixel boxes within 1e-5 relative + 2e-4 px; integer outputs exact."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import YOLOv1Head, synth

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_against_reference_fixtures(dev, golden):
    g = golden("g3_head.npz")
    head = YOLOv1Head()
    out = t(g["out"]).to(dev)
    torch.testing.assert_close(head.get_confidence(out).cpu(), t(g["conf"]), rtol=0, atol=1e-6)
    torch.testing.assert_close(head.get_classification_prob(out).cpu(), t(g["cls_prob"]), rtol=1e-5, atol=1e-7)
    assert torch.equal(head._get_classification_index(out).cpu(), t(g["cls_idx"]))
    assert torch.equal(head.get_confidence_mask(out, 0.5).cpu(), t(g["conf_mask_050"]))
    assert torch.equal(head.get_confidence_mask(out, 0.9).cpu(), t(g["conf_mask_090"]))


def test_single_image_fixture(dev, golden):
    """golden g2: the reference's per-image get_norm_xywh (sigmoid + cell order) feeds
    its own denormalize_xywh/xywh_to_ltrb once rearranged to (cell, predictor)."""
    g = golden("g2_decode.npz")
    head = YOLOv1Head()
    pred = t(g["pred_img"])[None].contiguous()
    ref_head = type("H", (), {})()
    norm = t(g["norm_boxmajor"])
    cellmajor = torch.stack([norm[:49], norm[49:]], dim=1).view(1, 49, 2, 4)
    want = ref_cpu.xywh_to_ltrb(ref_cpu.denormalize_xywh(cellmajor, 7, 448), 448)
    torch.testing.assert_close(head.get_ltrb(pred.to(dev)).cpu(), want, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("S,Bx,C,img,batch", [(7, 2, 20, 448, 33), (14, 5, 80, 448, 3), (3, 1, 4, 96, 5)])
def test_against_oracle(dev, S, Bx, C, img, batch):
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    out = synth.make_pred(batch, S, Bx, C, seed=S)
    d = out.to(dev)
    torch.testing.assert_close(head.get_ltrb(d).cpu(), ref_cpu.get_ltrb(out, S, Bx, img), rtol=1e-5, atol=2e-4)
    torch.testing.assert_close(head.get_confidence(d).cpu(), ref_cpu.get_confidence(out, Bx), rtol=0, atol=1e-6)
    torch.testing.assert_close(head.get_classification_prob(d).cpu(),
                               ref_cpu.get_classification_prob(out, Bx, C), rtol=1e-5, atol=1e-7)
    assert torch.equal(head._get_classification_index(d).cpu(), ref_cpu.get_classification_index(out, Bx, C))
    assert torch.equal(head.get_confidence_mask(d, 0.6).cpu(), ref_cpu.get_confidence_mask(out, 0.6, Bx))
    torch.testing.assert_close(head.get_max_confidence_ltrb(d).cpu(),
                               ref_cpu.get_max_confidence_ltrb(out, S, Bx, img), rtol=1e-5, atol=2e-4)


def test_output_shapes_follow_the_reference(dev):
    head = YOLOv1Head()
    d = synth.make_pred(2).to(dev)
    assert head.get_ltrb(d).shape == (2, 49, 2, 4)
    assert head.get_confidence(d).shape == (2, 49, 2, 1)                 # model.py:245
    assert head.get_classification_prob(d).shape == (2, 49, 1, 20)
    assert head._get_classification_index(d).shape == (2, 49, 1, 1)
    assert head.get_confidence_mask(d, 0.5).shape == (2, 49, 1, 1)
    assert head.get_confidence_mask(d, 0.5).dtype == torch.bool
    assert head.get_max_confidence_ltrb(d).shape == (2, 49, 1, 4)
