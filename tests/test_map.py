"""11-point mAP (eval.py:86-149): the oracle and the HIP kernels against values
produced by the reference's own get_map / get_ap (golden g8), bit for bit."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu


def cases(golden):
    g = golden("g8_map.npz")
    for k in "abcd":       # d: NaN confidences (ranked first, as torch.argsort does)
        yield (k, g[f"pred_{k}"], g[f"gt_{k}"], int(g[f"ncls_{k}"]), float(g[f"thr_{k}"]),
               float(g[f"map_{k}"]), g[f"ap_{k}"])


def test_oracle_map_equals_reference(golden):
    for k, pred, gt, ncls, thr, want_map, want_ap in cases(golden):
        assert ref_cpu.get_map(pred, gt, ncls, thr) == want_map, k
        got = np.array([ref_cpu.get_ap(pred, gt, c, thr) for c in range(ncls)])
        np.testing.assert_array_equal(got, want_ap)
    assert any(m > 0 for _, _, _, _, _, m, _ in cases(golden))


@pytest.mark.gpu
def test_gpu_map_equals_reference(dev, golden):
    from yolo_amd import box_ops

    for k, pred, gt, ncls, thr, want_map, want_ap in cases(golden):
        p, g = torch.from_numpy(pred).float().to(dev), torch.from_numpy(gt).float().to(dev)
        ap, m = box_ops._map(p, g, ncls, thr)
        np.testing.assert_array_equal(ap.cpu().numpy(), want_ap)        # float64, bit for bit
        assert m.item() == want_map, k
        assert box_ops.get_map(p, g, ncls, thr).item() == want_map
        assert box_ops.get_ap(p, g, 3, thr, ncls).item() == want_ap[3]


@pytest.mark.gpu
def test_gpu_map_edge_cases_and_pipeline(dev):
    from yolo_amd import MODE_REFERENCE, YOLOv1Head, box_ops, synth

    # no predictions / no ground truth -> 0, as the reference's arithmetic gives
    z6, z5 = torch.zeros(0, 6, device=dev), torch.zeros(0, 5, device=dev)
    one = torch.tensor([[0., 0., 10., 10., .9, 1.]], device=dev)
    assert box_ops.get_map(z6, torch.tensor([[0., 0., 10., 10., 1.]], device=dev), 3).item() == 0.0
    assert box_ops.get_map(one, z5, 3).item() == 0.0
    g1 = torch.tensor([[0., 0., 10., 10., 1.]])
    assert box_ops.get_map(one, g1.to(dev), 3).item() == ref_cpu.get_map(one.cpu(), g1, 3)
    # detections straight from the NMS kernel feed the metric (eval.py:89-94 row layouts)
    head = YOLOv1Head()
    pred = synth.make_pred(1, seed=5).to(dev)
    dets, count = head.postprocess(pred, 0.3, 0.5, MODE_REFERENCE)
    d = dets[0, : int(count[0])].contiguous()
    ltrb, label, cnt = synth.make_gt_boxes(1, seed=6)
    gt = torch.cat([ltrb[0, : int(cnt[0])], label[0, : int(cnt[0]), None].float()], dim=1)
    got = box_ops.get_map(d, gt.to(dev), 20, 0.5).item()
    assert got == ref_cpu.get_map(d.cpu(), gt, 20, 0.5)
    # larger random problem against the oracle
    gen = torch.Generator().manual_seed(3)
    P, G = 3000, 150
    gb = torch.rand(G, 2, generator=gen) * 300
    gtl = torch.cat([gb, gb + 20 + torch.rand(G, 2, generator=gen) * 100,
                     torch.randint(0, 20, (G, 1), generator=gen).float()], dim=1)
    src = torch.randint(0, G, (P,), generator=gen)
    pl = torch.cat([gtl[src, :4] + torch.randn(P, 4, generator=gen) * 8, torch.rand(P, 1, generator=gen),
                    torch.where(torch.rand(P, 1, generator=gen) < 0.6, gtl[src, 4:5],
                                torch.randint(0, 20, (P, 1), generator=gen).float())], dim=1)
    pl[10, 4] = pl[20, 4]                                        # a confidence tie
    assert box_ops.get_map(pl.to(dev), gtl.to(dev), 20, 4.0).item() == ref_cpu.get_map(pl, gtl, 20, 4.0)
    # a NaN confidence ranks first (before every number, NaNs among themselves by row)
    pn = pl.clone()
    pn[5, 4] = float("nan")
    pn[77, 4] = float("nan")
    assert box_ops.get_map(pn.to(dev), gtl.to(dev), 20, 4.0).item() == ref_cpu.get_map(pn, gtl, 20, 4.0)


def _random_problem(P, G, ncls, seed):
    gen = torch.Generator().manual_seed(seed)
    gb = torch.rand(G, 2, generator=gen) * 400
    gtl = torch.cat([gb, gb + 10 + torch.rand(G, 2, generator=gen) * 40,
                     torch.randint(0, ncls, (G, 1), generator=gen).float()], dim=1)
    src = torch.randint(0, G, (P,), generator=gen)
    pl = torch.cat([gtl[src, :4] + torch.randn(P, 4, generator=gen) * 6,
                    torch.rand(P, 1, generator=gen),
                    torch.where(torch.rand(P, 1, generator=gen) < 0.6, gtl[src, 4:5],
                                torch.randint(0, ncls, (P, 1), generator=gen).float())], dim=1)
    n_tie = min(pl[::97].shape[0], pl[1::97].shape[0])
    pl[::97][:n_tie, 4] = pl[1::97, 4][:n_tie]                   # confidence ties
    return pl, gtl


def _oracle_map_chunked(pl, gtl, ncls, thr, chunk=8192):
    """ref_cpu.get_map with the P x G IoU table built a slice of predictions at a time."""
    tp = np.concatenate([ref_cpu.true_positive_flags(pl[i:i + chunk], gtl, thr)
                         for i in range(0, pl.shape[0], chunk)])
    aps = np.array([ref_cpu.get_ap(pl, gtl, c, thr, tp) for c in range(ncls)])
    s = 0
    for a in aps:
        s += a
    return aps, s / ncls


@pytest.mark.gpu
def test_gpu_map_at_dataset_size(dev):
    """A VOC-val-sized call: 100,000 detections against 5,000 ground truths (sorted by
    class on the device), 20 classes -- float64 AP per class and mAP equal to the
    oracle's bit for bit; and the direct path (few ground truths) at the same P."""
    from yolo_amd import box_ops

    for P, G, thr in ((100_000, 5_000, 4.0), (100_000, 300, 2.0), (4097, 2049, 6.0)):
        pl, gtl = _random_problem(P, G, 20, seed=P + G)
        want_ap, want_map = _oracle_map_chunked(pl, gtl, 20, thr)
        ap, m = box_ops._map(pl.to(dev), gtl.to(dev), 20, thr)
        np.testing.assert_array_equal(ap.cpu().numpy(), want_ap)
        assert m.item() == want_map
        assert want_map > 1e-4                                    # some true positives in every regime


@pytest.mark.gpu
def test_gpu_map_small_calls(dev):
    """Calls of at most 1,024 predictions run in one workgroup (one launch): ragged sizes,
    ground truths in several LDS chunks, classes with far more than 64 true positives,
    more classes than wavefronts, confidence ties and NaN confidences -- equal to the
    oracle bit for bit, and to the four-kernel path on the same rows plus one."""
    from yolo_amd import box_ops

    for P, G, ncls, thr in ((1, 1, 3, 0.5), (63, 7, 20, 2.0), (1000, 300, 3, 1.0), (1024, 2048, 20, 4.0),
                            (1024, 32, 1000, 2.0), (777, 700, 40, 3.0)):
        pl, gtl = _random_problem(P, G, ncls, seed=7 * P + G)
        if P > 100:
            pl[5, 4] = float("nan")
            pl[77, 4] = float("nan")
            pl[9, 5] = 2.5                                        # not a class: belongs to no list
        want_ap, want_map = _oracle_map_chunked(pl, gtl, ncls, thr)
        ap, m = box_ops._map(pl.to(dev), gtl.to(dev), ncls, thr)
        np.testing.assert_array_equal(ap.cpu().numpy(), want_ap)
        assert m.item() == want_map, (P, G, ncls)
    # the same rows through the general path: one more prediction that changes nothing --
    # its confidence is the lowest and it is no true positive
    pl, gtl = _random_problem(1024, 200, 5, seed=11)
    extra = torch.tensor([[0., 0., 1., 1., -1.0, 0.]])
    a1, m1 = box_ops._map(pl.to(dev), gtl.to(dev), 5, 1.0)
    a2, m2 = box_ops._map(torch.cat([pl, extra]).to(dev), gtl.to(dev), 5, 1.0)
    assert torch.equal(a1, a2) and m1.item() == m2.item() and m1.item() > 0.01


@pytest.mark.gpu
def test_gpu_map_limits(dev):
    from yolo_amd import _lib
    lib = _lib.load()
    f = 0x1000
    assert lib.yh_map_11point(f, f, (1 << 24) + 1, 10, 20, 0.5, f, f, f, 1 << 30, None) == _lib.YH_ERR_UNSUPPORTED
    assert lib.yh_map_11point(f, f, 10, 10, 20, 0.5, f, f, f, 16, None) == _lib.YH_ERR_WORKSPACE
    assert lib.yh_map_workspace_bytes(1 << 20, 50_000) >= 5 * 4 * (1 << 20)
