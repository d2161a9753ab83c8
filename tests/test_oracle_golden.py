"""The CPU oracle against fixtures produced by the reference's own code
(oracle/make_golden.py).  Everything here is bit-exact: both sides are fp32
PyTorch on the CPU running the same operations in the same order."""
import numpy as np
import torch

from oracle import ref_cpu
from yolo_amd import synth, targets


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_g1_box_geometry(golden):
    g = golden("g1_iou.npz")
    b1, b2 = t(g["ltrb1"]), t(g["ltrb2"])
    assert torch.equal(ref_cpu.get_area(b1), t(g["area1"]))
    assert torch.equal(ref_cpu.get_area(b2), t(g["area2"]))
    assert torch.equal(ref_cpu.get_intersection_area(b1, b2), t(g["inter"]))
    iou = ref_cpu.get_iou(b1, b2)
    assert torch.equal(iou, t(g["iou"]))
    # elementwise form == the diagonal of the pairwise form (repair R3)
    assert torch.equal(ref_cpu.iou_elementwise(b1, b2), torch.diagonal(t(g["iou"])))
    # degenerate rows of the fixture
    assert iou[0, 0] == 0 and iou[5, 5] == 0        # zero area both sides: 0/0 -> 0
    assert iou[1, 1] == 0                           # inverted box has no area
    assert iou[2, 2] == 1 and iou[6, 6] == 1        # identical
    assert iou[3, 3] == 0                           # touching edges
    assert iou[4, 4] == 0.25                        # nested


def test_g1_numpy_iou_matches(golden):
    g = golden("g1_iou.npz")
    b1, b2 = g["ltrb1"], g["ltrb2"]
    for i in range(b1.shape[0]):
        np.testing.assert_array_equal(ref_cpu._iou_one_to_many(b1[i], b2), g["iou"][i])


def test_g2_decode(golden):
    g = golden("g2_decode.npz")
    pred = t(g["pred_img"])[None]                                   # (1,30,7,7)
    norm = ref_cpu.get_norm_xywh(pred, 2)[0]                        # (49,2,4)
    ref = t(g["norm_boxmajor"])                                     # rows 0-48 box 1, 49-97 box 2
    assert torch.equal(norm[:, 0], ref[:49])
    assert torch.equal(norm[:, 1], ref[49:])
    cm = t(g["norm_cellmajor"]).view(1, 49, 2, 4)
    den = ref_cpu.denormalize_xywh(cm, 7, 448)
    assert torch.equal(den.view(98, 4), t(g["denorm"]))
    assert torch.equal(ref_cpu.xywh_to_ltrb(den, 448).view(98, 4), t(g["ltrb"]))
    # the all-ones row of cell 0 and the documented one-sided clips
    assert t(g["ltrb"])[0].tolist() == [0.0, 0.0, 288.0, 288.0]
    assert torch.equal(ref_cpu.xywh_to_ltrb(t(g["extra_xywh"]), 448), t(g["extra_ltrb"]))
    assert t(g["extra_ltrb"])[0].tolist() == [0.0, 0.0, -40.0, -40.0]


def test_g2_grid_axis_convention(golden):
    """First spatial axis carries x (repair R2): cell i*S+j gets x_off = 64*i."""
    z = torch.zeros(1, 49, 2, 4)
    den = ref_cpu.denormalize_xywh(z, 7, 448)[0, :, 0]
    for cell in (0, 1, 7, 13, 48):
        assert den[cell, 0].item() == 64 * (cell // 7)
        assert den[cell, 1].item() == 64 * (cell % 7)


def test_g3_head_primitives(golden):
    g = golden("g3_head.npz")
    out = t(g["out"])
    assert torch.equal(ref_cpu.get_confidence(out, 2), t(g["conf"]))
    assert torch.equal(ref_cpu.get_classification_prob(out, 2, 20), t(g["cls_prob"]))
    assert torch.equal(ref_cpu.get_confidence_mask(out, 0.5), t(g["conf_mask_050"]))
    assert torch.equal(ref_cpu.get_confidence_mask(out, 0.9), t(g["conf_mask_090"]))
    assert torch.equal(ref_cpu.get_classification_index(out), t(g["cls_idx"]))


def test_g4_object_mask(golden):
    g = golden("g4_objmask.npz")
    gt, cnt, mask = t(g["gt_ltrb"]), t(g["count"]), t(g["mask"])
    label = torch.zeros(gt.shape[0], gt.shape[1], dtype=torch.long)
    _, _, om = targets.encode_padded(gt, label, cnt)
    assert torch.equal(om.view(-1, 7, 7), mask)      # [x_cell, y_cell] indexing


def test_g5_target_normalisation(golden):
    g = golden("g5_targets.npz")
    xywh = targets.ltrb_to_xywh(t(g["ltrb"]))
    assert torch.equal(xywh, t(g["xywh"]))
    assert torch.equal(targets.normalize_xywh(xywh, 64, 448), t(g["norm"]))


def test_g6_terms_through_reference_code(golden):
    g = golden("g6_terms.npz")
    pred, gx, gc, om = t(g["pred"]), t(g["gt_norm_xywh"]), t(g["gt_cls_prob"]), t(g["obj_mask"])
    coord, conf, cls, resp = ref_cpu.loss_terms(pred, gx, gc, om)
    assert torch.equal(resp, t(g["resp_mask"]))
    assert torch.equal(coord, t(g["coord"]))
    assert torch.equal(cls, t(g["cls"]))
    # the reference's own composition lines for the confidence term (model.py:265-276)
    # and for the total (model.py:311-317), run with the patches make_golden.py lists
    assert torch.equal(conf, t(g["conf"]))
    assert torch.equal(ref_cpu.get_loss(pred, gx, gc, om), t(g["loss"]))


def test_g6_inputs_are_the_seeded_synthetic_ones(golden):
    g = golden("g6_terms.npz")
    pred, gx, gc, om = synth.make_inputs(16, seed=0)
    assert torch.equal(pred, t(g["pred"]))
    assert torch.equal(gx, t(g["gt_norm_xywh"]))
    assert torch.equal(gc, t(g["gt_cls_prob"]))
    assert torch.equal(om, t(g["obj_mask"]))


def test_g7_oracle_regression(golden):
    g = golden("g7_oracle_cfg1.npz")
    pred, gx, gc, om = synth.make_inputs(16, seed=0)
    res = ref_cpu.loss_and_grad(pred, gx, gc, om)
    torch.testing.assert_close(res["total"], t(g["total"]), rtol=1e-6, atol=0)
    torch.testing.assert_close(res["grad"], t(g["grad"]), rtol=1e-6, atol=1e-9)
    dets, count, keep = ref_cpu.postprocess(pred, 0.5, 0.5, mode=0)
    np.testing.assert_array_equal(count, g["count_mode0"])
    np.testing.assert_array_equal(keep, g["keep_idx_mode0"])
    np.testing.assert_array_equal(dets, g["dets_mode0"])


def test_loss_fp32_vs_fp64_oracle():
    """The fp32 oracle against the same graph in fp64: bounds the oracle's own
    rounding, i.e. how tight a tolerance on the HIP kernel can be."""
    pred, gx, gc, om = synth.make_inputs(64, seed=3)
    a = ref_cpu.loss_and_grad(pred, gx, gc, om)
    b = ref_cpu.loss_and_grad(pred, gx, gc, om, dtype=torch.float64)
    assert abs(a["loss"].item() - b["loss"].item()) <= 1e-5 * abs(b["loss"].item())
    scale = b["grad"].abs().max().item()
    assert (a["grad"].double() - b["grad"]).abs().max().item() <= 1e-5 * scale


def test_responsibility_properties():
    pred, gx, gc, om = synth.make_inputs(32, seed=5)
    _, _, _, resp = ref_cpu.loss_terms(pred, gx, gc, om)
    per_cell = resp[..., 0].sum(dim=2)
    assert torch.equal(per_cell > 0, om[:, :, 0, 0])     # exactly the object cells
    assert int(per_cell.max()) == 1                       # one predictor per object


def test_nms_spec_small_cases():
    ltrb = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [100, 100, 110, 110], [0, 0, 10, 10]], np.float32)
    score = np.array([0.9, 0.8, 0.7, 0.95], np.float32)
    cls = np.array([1, 1, 1, 0])
    present = np.ones(4, bool)
    kept = ref_cpu.nms_image(ltrb, score, cls, present, 0.5, 0.5)
    assert kept.tolist() == [3, 0, 2]                     # class 0 first; 1 suppressed by 0
    # inclusive confidence threshold, strict IoU threshold
    kept = ref_cpu.nms_image(ltrb, score, cls, present, 0.8, 0.5)
    assert kept.tolist() == [3, 0]
    same = np.array([[0, 0, 10, 10], [0, 0, 10, 20]], np.float32)   # IoU exactly 0.5
    kept = ref_cpu.nms_image(same, np.array([0.9, 0.8], np.float32), np.zeros(2, int), np.ones(2, bool), 0.0, 0.5)
    assert kept.tolist() == [0, 1]
    # equal scores: the lower flat index goes first and wins
    kept = ref_cpu.nms_image(ltrb[:2], np.array([0.5, 0.5], np.float32), np.zeros(2, int), np.ones(2, bool), 0.0, 0.5)
    assert kept.tolist() == [0]
    # nothing valid
    assert ref_cpu.nms_image(ltrb, score, cls, np.zeros(4, bool), 0.0, 0.5).size == 0
