"""Target encoding (the step in front of the loss): oracle against the reference's
fixtures on the CPU, and the HIP encoder bit-exact against the oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import synth, targets


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_oracle_encoder_against_reference_fixtures(golden):
    g4 = golden("g4_objmask.npz")
    gt, cnt = t(g4["gt_ltrb"]), t(g4["count"])
    lab = torch.zeros(gt.shape[0], gt.shape[1], dtype=torch.long)
    _, _, om = ref_cpu.encode_targets(gt, lab, cnt)
    assert torch.equal(om.view(-1, 7, 7), t(g4["mask"]))             # model.py:213-222
    g5 = golden("g5_targets.npz")
    boxes = t(g5["ltrb"])                                            # 64 boxes, one image each
    gx, _, om = ref_cpu.encode_targets(boxes[:, None, :], torch.zeros(64, 1, dtype=torch.long),
                                       torch.ones(64, dtype=torch.long))
    norm = t(g5["norm"])
    for b in range(64):
        cell = int(om[b, :, 0, 0].nonzero()[0, 0])
        assert torch.equal(gx[b, cell, 0], norm[b])                  # data.py:185-194


def test_host_encoder_equals_oracle():
    for S, C, img in ((7, 20, 448), (14, 80, 448), (3, 4, 96), (5, 7, 455)):
        ltrb, label, count = synth.make_gt_boxes(33, C, img, max_boxes=9, seed=S)
        a = ref_cpu.encode_targets(ltrb, label, count, S, C, img)
        b = targets.encode_padded(ltrb, label, count, S, C, img)
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    ragged = [ltrb[i, : int(count[i])] for i in range(33)]
    labs = [label[i, : int(count[i])] for i in range(33)]
    c = targets.encode_targets(ragged, labs, 5, 7, 455)
    assert all(torch.equal(x, y) for x, y in zip(a, c))
    with pytest.raises(ValueError):
        targets.encode_targets(ragged, [x + 100 for x in labs], 5, 7, 455)


def test_last_box_in_a_cell_wins():
    ltrb = torch.tensor([[[10., 10., 30., 30.], [12., 12., 40., 40.], [300., 300., 400., 420.]]])
    label = torch.tensor([[3, 5, 7]])
    gx, gc, om = ref_cpu.encode_targets(ltrb, label, torch.tensor([3]))
    assert int(om.sum()) == 2
    assert gc[0, 0, 0].argmax().item() == 5 and gc[0, 0, 0].sum().item() == 1.0
    assert gx[0, 0, 0].tolist() == [26 / 64, 26 / 64, 28 / 448, 28 / 448]
    cell = (350 // 64) * 7 + 360 // 64
    assert bool(om[0, cell, 0, 0]) and gc[0, cell, 0].argmax().item() == 7


@pytest.mark.gpu
@pytest.mark.parametrize("S,C,img,batch,mb", [(7, 20, 448, 257, 5), (14, 80, 448, 33, 12),
                                               (3, 4, 96, 9, 3), (5, 7, 455, 17, 20), (7, 20, 448, 6, 70),
                                               (7, 21, 448, 5, 1),
                                               # small grids with many boxes per image: the tiled
                                               # kernel's LDS would not fit, the general one runs
                                               (1, 5, 64, 300, 64), (2, 3, 64, 70, 64), (1, 2, 32, 1, 27),
                                               (2, 4, 64, 65, 33)])
def test_gpu_encoder_bit_exact(dev, S, C, img, batch, mb):
    from yolo_amd import YOLOv1Head

    head = YOLOv1Head(img_size=img, n_cells=S, n_classes=C)
    ltrb, label, count = synth.make_gt_boxes(batch, C, img, max_boxes=mb, seed=batch)
    ltrb[0, 0] = torch.tensor([0., 0., 0., 0.])                       # corner, zero size
    if batch > 1:
        ltrb[1, 0] = torch.tensor([img - 1., img - 1., img - 1., img - 1.])
    gx, gc, om = head.encode_targets(ltrb.to(dev), label.to(dev), count.to(dev))
    wx, wc, wm = ref_cpu.encode_targets(ltrb, label, count, S, C, img)
    assert torch.equal(om.cpu(), wm)
    assert torch.equal(gx.cpu().view(torch.int32), wx.view(torch.int32))     # bit for bit
    assert torch.equal(gc.cpu(), wc)


@pytest.mark.gpu
def test_gpu_encoder_edge_cases_and_pipeline(dev):
    from yolo_amd import YOLOv1Head

    head = YOLOv1Head()
    # no boxes at all, out-of-range label, padding beyond count is ignored
    ltrb = torch.rand(4, 3, 4) * 400
    ltrb[..., 2:] += ltrb[..., :2]
    ltrb = ltrb.clamp(max=447)
    label = torch.tensor([[1, 2, 3], [25, -1, 4], [0, 0, 0], [19, 19, 19]])
    count = torch.tensor([0, 3, 1, 2])
    gx, gc, om = head.encode_targets(ltrb.to(dev), label.to(dev), count.to(dev))
    wx, wc, wm = ref_cpu.encode_targets(ltrb, label, count)
    assert torch.equal(om.cpu(), wm) and torch.equal(gx.cpu(), wx) and torch.equal(gc.cpu(), wc)
    assert int(om[0].sum()) == 0 and float(gc[0].sum()) == 0.0
    # full size: encoder output feeds the loss kernel directly
    lt, lb, ct = synth.make_gt_boxes(4096, seed=1)
    gx, gc, om = head.encode_targets(lt.to(dev), lb.to(dev), ct.to(dev))
    _, hx, hc, hm = synth.make_inputs(4096, seed=0)                   # host-encoded, same boxes
    assert torch.equal(gx.cpu(), hx) and torch.equal(gc.cpu(), hc) and torch.equal(om.cpu(), hm)
    pred = synth.make_pred(4096, seed=0).to(dev)
    t1, g1 = head.loss_fwd_bwd(pred, gx, gc, om)
    t2, g2 = head.loss_fwd_bwd(pred, hx.to(dev), hc.to(dev), hm.to(dev))
    assert torch.equal(t1, t2) and torch.equal(g1, g2)
    e = head.encode_targets(torch.zeros(0, 3, 4, device=dev), torch.zeros(0, 3, dtype=torch.int32, device=dev),
                            torch.zeros(0, dtype=torch.int32, device=dev))
    assert e[0].shape == (0, 49, 1, 4)
