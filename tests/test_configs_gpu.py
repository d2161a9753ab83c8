"""The two large benchmark configurations at their full sizes, through properties
that do not need the CPU oracle on every image: the data-parallel case
(batch 65,536 split into 2/4/8 shards) and the NMS stress shape
(S=14, B=5, C=80, 980 candidates per image, conf 0.001, dense mode)."""
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import MODE_DENSE, YOLOv1Head, synth

pytestmark = pytest.mark.gpu


def test_batch_65536_shards_compose(dev):
    """What each of N ranks computes on its slice, run here one slice after the
    other: gradients are bit for bit the slices of the unsharded gradient, the
    term sums add up to the unsharded sums, detections are the unsharded rows."""
    from yolo_amd.dist import shard_range
    head = YOLOv1Head()
    B = 65536
    # 65,536 images as 16 differently seeded blocks of the synthetic generator
    parts = [synth.make_inputs(4096, seed=500 + 2 * i) for i in range(16)]
    pred, gx, gc, om = (torch.cat([p[j] for p in parts]).to(dev) for j in range(4))
    terms, grad = head.loss_fwd_bwd(pred, gx, gc, om)
    dets, count = head.postprocess(pred, 0.5, 0.5)
    assert torch.isfinite(terms).all() and torch.isfinite(grad).all()
    # a slice against the oracle (fp64), so that the big launch itself is anchored
    sl = slice(30000, 30064)
    ref = ref_cpu.loss_and_grad(pred[sl].cpu(), gx[sl].cpu(), gc[sl].cpu(), om[sl].cpu(),
                                dtype=torch.float64)
    scale = ref["grad"].abs().max().item()
    assert ((grad[sl].cpu().double() * (B / 64)) - ref["grad"]).abs().max().item() <= 1e-5 * scale
    for world in (2, 4, 8):
        sums = torch.zeros(4, dtype=torch.float64)
        for rank in range(world):
            lo, hi = shard_range(B, rank, world)
            t, g = head.loss_fwd_bwd(pred[lo:hi], gx[lo:hi], gc[lo:hi], om[lo:hi], global_batch=B)
            assert torch.equal(g, grad[lo:hi])
            sums += t[:4].cpu().double()
            d, c = head.postprocess(pred[lo:hi], 0.5, 0.5)
            assert torch.equal(d, dets[lo:hi]) and torch.equal(c, count[lo:hi])
        want = terms[:4].cpu().double()
        assert ((sums - want).abs() <= 1e-6 * want.abs()).all(), (sums, want)


def test_stress_shape_full_batch(dev):
    S, Bx, C, B = 14, 5, 80, 1024
    head = YOLOv1Head(img_size=448, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_pred(B, S, Bx, C, seed=77).to(dev)
    ct, it = 0.001, 0.5
    dets, count, kidx = head.postprocess(pred, ct, it, MODE_DENSE, return_idx=True)
    d2, c2, k2 = head.postprocess(pred, ct, it, MODE_DENSE, return_idx=True)
    assert torch.equal(dets, d2) and torch.equal(count, c2) and torch.equal(kidx, k2)
    ltrb, score, cls, present = head.decode_candidates(pred, MODE_DENSE)
    n = dets.shape[1]
    assert n == S * S * Bx == 980
    # two images against the oracle, bit-exact
    sl = slice(511, 513)
    want = ref_cpu.nms_batch(ltrb[sl].cpu().numpy(), score[sl].cpu().numpy(), cls[sl].cpu().numpy(),
                             present[sl].cpu().numpy().astype(bool), ct, it)
    assert (dets[sl].cpu().numpy() == want[0]).all()
    assert (count[sl].cpu().numpy() == want[1]).all()
    assert (kidx[sl].cpu().numpy() == want[2]).all()
    # properties over all 1,024 images
    live = torch.arange(n, device=dev)[None, :] < count[:, None]
    assert torch.equal(live, kidx >= 0) and (dets[~live] == 0).all()
    assert (dets[..., 4][live] >= ct).all()
    c, s = dets[..., 5], dets[..., 4]
    pair = live[:, 1:] & live[:, :-1]
    ok = (c[:, 1:] > c[:, :-1]) | ((c[:, 1:] == c[:, :-1]) & (
        (s[:, 1:] < s[:, :-1]) | ((s[:, 1:] == s[:, :-1]) & (kidx[:, 1:] > kidx[:, :-1]))))
    assert ok[pair].all()
    k = kidx.clamp(min=0).long()
    assert torch.equal(torch.gather(score, 1, k)[live], s[live])
    assert torch.equal(torch.gather(cls, 1, k)[live].float(), c[live])
    # idempotence: NMS over the survivors keeps every one of them where it is
    d3, c3 = YOLOv1Head.nms(dets[..., :4].contiguous(), s.contiguous(), c.to(torch.int32), live, ct, it)
    assert torch.equal(c3, count) and torch.equal(d3, dets)
    # every suppressed candidate has a kept one of its class with a higher-or-equal
    # score and IoU above the threshold (checked on a sample of images)
    for i in (0, 300, 1023):
        kept = kidx[i, :count[i]].long()
        dead = torch.ones(n, dtype=torch.bool, device=dev)
        dead[kept] = False
        dead &= score[i] >= ct
        iou = ref_cpu.get_iou(ltrb[i][dead], ltrb[i][kept])
        allowed = (cls[i][dead][:, None] == cls[i][kept][None, :]) & \
                  (score[i][kept][None, :] >= score[i][dead][:, None])
        assert ((iou > it) & allowed).any(dim=1).all()
    # the loss at the same shape and batch: shards compose here as well
    _, gx, gc, om = (x.to(dev) for x in synth.make_inputs(B, S, Bx, C, 448, seed=78))
    terms, grad = head.loss_fwd_bwd(pred, gx, gc, om)
    t0, g0 = head.loss_fwd_bwd(pred[:500], gx[:500], gc[:500], om[:500], global_batch=B)
    t1, g1 = head.loss_fwd_bwd(pred[500:], gx[500:], gc[500:], om[500:], global_batch=B)
    assert torch.equal(torch.cat([g0, g1]), grad)
    want = terms[:4].cpu().double()
    assert (((t0[:4] + t1[:4]).cpu().double() - want).abs() <= 1e-6 * want.abs()).all()


def test_maximum_sizes(dev):
    """The limits the header states: S = 64, Bx = 16, C = 4096 for the loss, the grid
    decode and -- through the workspace entry points -- decode + NMS; S*S*Bx = 2,048 rows
    per image for the NMS entry points without a workspace."""
    from yolo_amd import MODE_REFERENCE, _lib
    import numpy as np
    S, Bx, C = 64, 16, 4096
    head = YOLOv1Head(img_size=64 * 8, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred, gx, gc, om = synth.make_inputs(1, S, Bx, C, 64 * 8, seed=5)
    ref = ref_cpu.loss_and_grad(pred, gx, gc, om, n_cells=S, n_bboxes=Bx, n_classes=C, img_size=64 * 8)
    terms, grad = head.loss_fwd_bwd(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev))
    assert abs(terms[4].item() - ref["loss"].item()) <= 1e-5 * abs(ref["loss"].item())
    assert (grad.cpu() - ref["grad"]).abs().max().item() <= 1e-5 * ref["grad"].abs().max().item()
    ltrb = head.get_ltrb(pred.to(dev))
    want = ref_cpu.get_ltrb(pred, S, Bx, 64 * 8)
    torch.testing.assert_close(ltrb.cpu(), want, rtol=1e-5, atol=2e-4)
    # the entry point without a workspace stops at 2,048 candidates per image ...
    rc = _lib.load().yh_decode_nms(pred.to(dev).data_ptr(), 1, 1, None, 0, 0.5, 0.5, 1, S, Bx, C, 512, None)
    assert rc == _lib.YH_ERR_UNSUPPORTED
    # ... postprocess goes on through yh_decode_nms_ws: the largest shape the loss takes
    # (65,536 candidates per image) in both candidate modes
    for mode in (MODE_DENSE, MODE_REFERENCE):
        got = head.postprocess(pred.to(dev), 0.3, 0.5, mode, return_idx=True)
        l, s, c, p = head.decode_candidates(pred.to(dev), mode)
        want = ref_cpu.nms_batch(l.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), p.cpu().numpy(), 0.3, 0.5)
        for g_, w in zip(got, want):
            np.testing.assert_array_equal(g_.cpu().numpy().view(np.uint32), np.asarray(w).view(np.uint32))
        assert int(got[1].sum()) > 1000
    for S, Bx, C, mode in ((32, 1, 4096, MODE_DENSE), (8, 16, 33, MODE_DENSE), (22, 2, 7, MODE_REFERENCE),
                           (32, 2, 4096, MODE_DENSE), (32, 2, 7, MODE_REFERENCE), (45, 1, 11, MODE_DENSE),
                           # above 2,048 candidates per image: few classes (long chains against
                           # the kept rows), many classes, a bias folded in
                           (64, 2, 20, MODE_DENSE), (64, 2, 3, MODE_REFERENCE), (33, 2, 300, MODE_DENSE),
                           (23, 4, 5, MODE_DENSE), (64, 16, 20, MODE_DENSE)):
        head = YOLOv1Head(img_size=32 * S, n_cells=S, n_bboxes=Bx, n_classes=C)
        pred = synth.make_pred(2, S, Bx, C, seed=S + Bx).to(dev)
        got = head.postprocess(pred, 0.3, 0.5, mode, return_idx=True)
        l, s, c, p = head.decode_candidates(pred, mode)
        want = ref_cpu.nms_batch(l.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), p.cpu().numpy(), 0.3, 0.5)
        for g_, w in zip(got, want):
            np.testing.assert_array_equal(g_.cpu().numpy().view(np.uint32), np.asarray(w).view(np.uint32))
