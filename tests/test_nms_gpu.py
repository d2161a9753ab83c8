"""HIP decode + NMS against the CPU oracle, through the C ABI.

Bit-exact where the domain is index/compare work: the NMS keep sets, their
order, the copied box/score bits and the counts.  The decode in front of it is
floating point (sigmoid): GPU and CPU exponentials differ in the last ulp, so
the decode is compared at a tolerance (boxes: 1e-5 relative + 2e-4 px absolute,
scores 1e-6 absolute) and the NMS parity tests feed the SAME decoded candidates
to both sides.  Documented tie cases (results may legitimately differ between
CPU and GPU decode): two predictors of a cell whose confidences round equal,
class logits whose softmax values round equal, a score that lands exactly on
the confidence threshold, IoU exactly on the IoU threshold.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import MODE_DENSE, MODE_REFERENCE, YOLOv1Head, synth

pytestmark = pytest.mark.gpu


def oracle_nms(ltrb, score, cls, present, ct, it):
    return ref_cpu.nms_batch(ltrb.cpu().numpy(), score.cpu().numpy(), cls.cpu().numpy(),
                             present.cpu().numpy(), ct, it)


def assert_same(got, want):
    dets, count, kidx = (x.cpu().numpy() for x in got)
    wd, wc, wk = want
    np.testing.assert_array_equal(count, wc)
    np.testing.assert_array_equal(kidx, wk)
    np.testing.assert_array_equal(dets.view(np.uint32), wd.view(np.uint32))   # bit for bit


def random_candidates(batch, n, n_classes, seed, img=448.0, max_wh=200.0):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(batch, n, 2, generator=g) * img
    wh = torch.rand(batch, n, 2, generator=g) * max_wh + 4
    ltrb = torch.cat([(c - wh / 2).clamp(min=0), (c + wh / 2).clamp(max=img)], dim=-1)
    score = torch.rand(batch, n, generator=g)
    cls = torch.randint(0, n_classes, (batch, n), generator=g, dtype=torch.int32)
    present = torch.rand(batch, n, generator=g) > 0.1
    return ltrb.contiguous(), score, cls, present


# --------------------------------------------------------------------------- decode
@pytest.mark.parametrize("mode", [MODE_REFERENCE, MODE_DENSE])
def test_decode_candidates_match_oracle(dev, mode):
    head = YOLOv1Head()
    pred = synth.make_pred(64, seed=0)
    ltrb, score, cls, present = head.decode_candidates(pred.to(dev), mode)
    o_ltrb, o_score, o_cls, o_present = ref_cpu.decode_candidates(pred, mode)
    torch.testing.assert_close(ltrb.cpu(), o_ltrb, rtol=1e-5, atol=2e-4)
    torch.testing.assert_close(score.cpu(), o_score, rtol=0, atol=1e-6)
    assert torch.equal(cls.cpu().long(), o_cls)
    assert torch.equal(present.cpu(), o_present)


# ------------------------------------------------------- NMS proper, same inputs both sides
@pytest.mark.parametrize("batch,n,ncls,ct,it", [
    (8, 49, 20, 0.5, 0.5),      # one wavefront per image, one candidate per lane
    (8, 98, 20, 0.3, 0.5),      # two candidates per lane
    (5, 1, 3, 0.0, 0.5),
    (4, 64, 1, 0.0, 0.3),       # a single class: the longest possible segments
    (4, 128, 2, 0.0, 0.4),
    (6, 64, 64, 0.0, 0.4),      # classes 0..63: the largest table for candidates handed over in memory
    (6, 64, 65, 0.3, 0.4),      # a class of 64 or more among the valid ones: the ranked path
    (6, 100, 64, 0.0, 0.4),
    (6, 100, 70, 0.2, 0.4),
    (3, 200, 5, 0.2, 0.5),      # one workgroup per image
    (3, 500, 80, 0.001, 0.5),
    (2, 980, 80, 0.001, 0.5),   # the stress shape S=14, B=5, C=80
    (2, 1024, 1, 0.0, 0.6),     # one class: the step loop over a 1,024-member segment
    (2, 1500, 40, 0.001, 0.5),  # eight candidates per thread
    (2, 2048, 300, 0.0, 0.5),   # maximum size, short segments (pair path)
    (1, 2048, 3, 0.0, 0.4),     # maximum size, segments longer than 64 (step loop)
    (3, 2049, 20, 0.1, 0.5),    # the first size of the global-memory path (yh_nms_ws)
    (2, 5000, 1, 0.0, 0.5),     # one class: every chunk runs against all rows kept so far
    (2, 8192, 4096, 0.0, 0.5),  # more classes than members per class
    (1, 65536, 20, 0.2, 0.45),  # the largest size
])
def test_nms_bit_exact(dev, batch, n, ncls, ct, it):
    ltrb, score, cls, present = random_candidates(batch, n, ncls, seed=n + batch)
    got = YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls.to(dev), present.to(dev), ct, it,
                         return_idx=True)
    assert_same(got, oracle_nms(ltrb, score, cls, present, ct, it))


def test_class_segments_around_the_block_length(dev):
    """A class segment is worked on in blocks of 64 members, a round per block: segments of
    1, 63, 64, 65, 127, 128, 129, 192 and 193 members side by side in one image, overlapping
    heavily, so that survivors of an early block must stop members of the later ones."""
    lens = [1, 63, 64, 65, 127, 128, 129, 192, 193]
    n = sum(lens)                                        # 962 candidates
    g = torch.Generator().manual_seed(11)
    cls = torch.cat([torch.full((m,), c, dtype=torch.int32) for c, m in enumerate(lens)])
    batch = 3
    c = torch.rand(batch, n, 2, generator=g) * 120 + 100      # crowded: many IoUs above 0.3
    wh = torch.rand(batch, n, 2, generator=g) * 60 + 30
    ltrb = torch.cat([c - wh / 2, c + wh / 2], dim=-1).contiguous()
    score = torch.rand(batch, n, generator=g)
    score[1] = (score[1] * 8).floor() / 8                     # equal scores: order by index
    perm = torch.randperm(n, generator=g)
    cls_b = cls[perm].unsqueeze(0).expand(batch, n).contiguous()
    present = torch.ones(batch, n, dtype=torch.bool)
    for it in (0.3, 0.5, 0.05):
        got = YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls_b.to(dev), present.to(dev), 0.0, it,
                             return_idx=True)
        assert_same(got, oracle_nms(ltrb, score, cls_b, present, 0.0, it))


@pytest.mark.parametrize("S,Bx,C", [(14, 5, 80), (13, 5, 20), (10, 5, 3), (20, 5, 80)])
def test_one_class_takes_most_cells(dev, S, Bx, C):
    """Detections of a real image crowd into few classes: one class wins 70 % of the cells
    (segments of several hundred members), through the fused kernels."""
    img = 32 * S
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_pred(4, S, Bx, C, seed=S + C)
    g = torch.Generator().manual_seed(S)
    crowd = torch.rand(4, S, S, generator=g) < 0.7
    pred[:, 5 * Bx + C // 2][crowd] += 30.0
    pred = pred.to(dev)
    for ct, it in ((0.001, 0.5), (0.2, 0.3)):
        got = head.postprocess(pred, ct, it, MODE_DENSE, return_idx=True)
        ltrb, score, cls, present = head.decode_candidates(pred, MODE_DENSE)
        assert int((cls[0] == C // 2).sum()) > 64
        assert_same(got, oracle_nms(ltrb, score, cls, present, ct, it))


@pytest.mark.parametrize("cells_a,cells_b", [(38, 39), (51, 52), (13, 64), (39, 103)])
def test_class_buckets_around_the_ranking_switch(dev, cells_a, cells_b):
    """The fused kernel ranks a class bucket by counting up to 192 candidates and by sorted
    chunks of 64 plus binary searches above: two classes win exact numbers of cells (5
    candidates each: 190 and 195, 255 and 260, 65 and 320, 195 and 515 candidates), chunks
    that end on and next to a multiple of 64, in one image with equal scores."""
    S, Bx, C = 14, 5, 80
    head = YOLOv1Head(img_size=448, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_pred(3, S, Bx, C, seed=cells_a * 1000 + cells_b)
    g = torch.Generator().manual_seed(cells_a + cells_b)
    for b in range(3):
        order = torch.randperm(S * S, generator=g)
        flat = pred[b].view(5 * Bx + C, S * S)
        flat[5 * Bx + 7] -= 40.0                  # these two classes win the chosen cells only
        flat[5 * Bx + 61] -= 40.0
        flat[5 * Bx + 7, order[:cells_a]] += 80.0
        flat[5 * Bx + 61, order[cells_a:cells_a + cells_b]] += 80.0
    # image 1: confidences on a coarse grid, so that many candidates tie and the index decides
    conf = pred[1, 4:5 * Bx:5]
    pred[1, 4:5 * Bx:5] = (conf * 2).round() / 2
    pred = pred.to(dev)
    ltrb, score, cls, present = head.decode_candidates(pred, MODE_DENSE)
    assert int((cls[0] == 7).sum()) == 5 * cells_a and int((cls[0] == 61).sum()) == 5 * cells_b
    for ct, it in ((0.001, 0.5), (0.3, 0.2)):
        got = head.postprocess(pred, ct, it, MODE_DENSE, return_idx=True)
        assert_same(got, oracle_nms(ltrb, score, cls, present, ct, it))


@pytest.mark.parametrize("size", [1e-19, 3e-16, 1e-12])
def test_boxes_whose_areas_are_denormal(dev, size):
    """Boxes of 1e-19 px have areas of 1e-38, among the denormals, where a product or a
    reciprocal no longer carries 24 bits: such pairs must take the division."""
    g = torch.Generator().manual_seed(5)
    batch, n = 2, 300
    lt = torch.rand(batch, n, 2, generator=g) * (2 * size)
    wh = torch.rand(batch, n, 2, generator=g) * (2 * size) + size
    ltrb = torch.cat([lt, lt + wh], dim=-1).contiguous()
    score = torch.rand(batch, n, generator=g)
    cls = torch.randint(0, 3, (batch, n), generator=g, dtype=torch.int32)
    present = torch.ones(batch, n, dtype=torch.bool)
    for it in (0.3, 0.5):
        got = YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls.to(dev), present.to(dev), 0.0, it,
                             return_idx=True)
        want = oracle_nms(ltrb, score, cls, present, 0.0, it)
        assert 3 < want[1].min() and want[1].max() < n      # some suppressed, some kept
        assert_same(got, want)


def test_iou_threshold_on_and_next_to_a_pair_iou(dev):
    """The workgroup-per-image kernel decides IoU > threshold without dividing;
    thresholds that ARE the IoU of some pair, and the floats next to it on either
    side (odd and even last bits, so both tie directions), must give the oracle's
    keep sets bit for bit.  Also thresholds outside [0, 1e30), which divide."""
    ltrb, score, cls, present = random_candidates(2, 300, 30, seed=77)
    b = ltrb[0].numpy()
    c = cls[0].numpy()
    thresholds = []
    for i in range(300):
        for j in range(i + 1, 300):
            if c[i] != c[j]:
                continue
            iw = max(np.float32(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0])), np.float32(0))
            ih = max(np.float32(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1])), np.float32(0))
            inter = np.float32(iw * ih)
            a1 = np.float32(max(b[i, 2] - b[i, 0], 0)) * np.float32(max(b[i, 3] - b[i, 1], 0))
            a2 = np.float32(max(b[j, 2] - b[j, 0], 0)) * np.float32(max(b[j, 3] - b[j, 1], 0))
            uni = np.float32(np.float32(a1 + a2) - inter)
            if inter > 0 and uni > 0:
                thresholds.append(np.float32(inter / uni))
        if len(thresholds) >= 12:
            break
    assert len(thresholds) >= 12
    cases = []
    for v in thresholds[:12]:
        cases += [float(np.nextafter(v, np.float32(-1))), float(v), float(np.nextafter(v, np.float32(2)))]
    cases += [0.0, -0.25, 2e30, float("inf"), float(np.float32(1e-45))]
    for it in cases:
        got = YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls.to(dev), present.to(dev), 0.0, it,
                             return_idx=True)
        assert_same(got, oracle_nms(ltrb, score, cls, present, 0.0, it))


def test_nms_tie_and_degenerate_cases(dev):
    f = torch.tensor
    ltrb = f([[[0, 0, 10, 10], [0, 0, 10, 20], [0, 0, 10, 10], [5, 5, 5, 5], [30, 30, 20, 20],
               [100, 100, 120, 120], [100, 100, 120, 120], [0, 0, 448, 448]]], dtype=torch.float32)
    score = f([[0.5, 0.5, 0.5, 0.9, 0.8, 0.7, 0.7, 0.5]], dtype=torch.float32)
    cls = torch.zeros(1, 8, dtype=torch.int32)
    present = torch.ones(1, 8, dtype=torch.bool)
    for ct, it in ((0.0, 0.5), (0.5, 0.5), (0.5000001, 0.5), (0.0, 0.0), (0.0, 1.0), (0.7, 0.49999)):
        got = YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls.to(dev), present.to(dev), ct, it,
                             return_idx=True)
        assert_same(got, oracle_nms(ltrb, score, cls, present, ct, it))
    # IoU of rows 0 and 1 is exactly 0.5: strict '>' keeps both at iou_thresh 0.5
    got = YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls.to(dev), present.to(dev), 0.0, 0.5,
                         return_idx=True)
    kept = got[2][0, : int(got[1][0])].cpu().tolist()
    assert 0 in kept and 1 in kept and 2 not in kept      # equal scores: lower index wins
    # nothing present / nothing above the threshold / negative scores and -0.0
    none = torch.zeros(1, 8, dtype=torch.bool)
    got = YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls.to(dev), none.to(dev), 0.0, 0.5, return_idx=True)
    assert_same(got, oracle_nms(ltrb, score, cls, none, 0.0, 0.5))
    neg = f([[-1.0, -0.0, 0.0, -2.0, 1.0, -0.5, 0.25, -0.0]], dtype=torch.float32)
    got = YOLOv1Head.nms(ltrb.to(dev), neg.to(dev), cls.to(dev), present.to(dev), -5.0, 0.5, return_idx=True)
    assert_same(got, oracle_nms(ltrb, neg, cls, present, -5.0, 0.5))
    # present=None means all present
    got = YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls.to(dev), None, 0.0, 0.5, return_idx=True)
    assert_same(got, oracle_nms(ltrb, score, cls, present, 0.0, 0.5))


def test_reference_mode_thresholds_and_ragged_batches(dev):
    """Reference mode of the benchmark shape at thresholds that keep few, about half
    and nearly all cells, over odd and even batch sizes (the last workgroup is then
    partly empty), with and without a bias."""
    head = YOLOv1Head()
    for batch in (1, 2, 7, 8, 9, 33):
        pred = synth.make_pred(batch, seed=100 + batch).to(dev)
        ltrb, score, cls, present = head.decode_candidates(pred, MODE_REFERENCE)
        for ct in (0.9, 0.5, 0.35, 0.05, 0.0):
            got = head.postprocess(pred, ct, 0.4, MODE_REFERENCE, return_idx=True)
            assert_same(got, oracle_nms(ltrb, score, cls, present, ct, 0.4))
    bias = torch.randn(1470, device=dev)
    pred = synth.make_pred(9, seed=5).to(dev)
    z = pred - bias.view(1, 30, 7, 7)
    for ct in (0.5, 0.05):
        want = head.postprocess(z + bias.view(1, 30, 7, 7), ct, 0.5, MODE_REFERENCE, return_idx=True)
        got = head.postprocess(z, ct, 0.5, MODE_REFERENCE, return_idx=True, bias=bias)
        for g_, w_ in zip(got, want):
            assert torch.equal(g_, w_)


# ------------------------------------------------ fused kernel == NMS of its own candidates
@pytest.mark.parametrize("S,Bx,C,batch,mode,ct,it", [
    (7, 2, 20, 16, MODE_REFERENCE, 0.5, 0.5),
    (7, 2, 20, 16, MODE_DENSE, 0.1, 0.3),
    (7, 2, 20, 37, MODE_DENSE, 0.5, 0.5),
    (14, 5, 80, 4, MODE_DENSE, 0.001, 0.5),      # NMS stress configuration
    (14, 5, 80, 4, MODE_REFERENCE, 0.3, 0.5),
    (3, 1, 4, 9, MODE_REFERENCE, 0.2, 0.5),
    (11, 3, 7, 5, MODE_DENSE, 0.4, 0.45),
    (20, 5, 80, 2, MODE_DENSE, 0.001, 0.5),      # 2,000 candidates per image
    (40, 1, 6, 2, MODE_REFERENCE, 0.3, 0.5),     # 1,600 cells
])
def test_fused_decode_nms(dev, S, Bx, C, batch, mode, ct, it):
    img = 32 * S
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_pred(batch, S, Bx, C, seed=S * 100 + Bx).to(dev)
    got = head.postprocess(pred, ct, it, mode, return_idx=True)
    ltrb, score, cls, present = head.decode_candidates(pred, mode)
    assert_same(got, oracle_nms(ltrb, score, cls, present, ct, it))
    # and the stand-alone NMS entry point on the same candidates
    got2 = YOLOv1Head.nms(ltrb, score, cls, present, ct, it, return_idx=True)
    assert_same(got2, oracle_nms(ltrb, score, cls, present, ct, it))


@pytest.mark.parametrize("S,Bx,C,mode", [
    (7, 2, 20, MODE_DENSE),      # the headline shape with a candidate per predictor (98 slots)
    (8, 2, 64, MODE_DENSE),      # all 128 slots, the largest class table inside the key array
    (8, 2, 65, MODE_DENSE),      # one class more: the second half's mask words move to the histogram arrays
    (8, 2, 128, MODE_DENSE),     # the largest class table without a sort
    (8, 2, 129, MODE_DENSE),     # one class more: the ranked path
    (8, 2, 1, MODE_DENSE),       # one class of up to 128 members: the masks' second word, long kill chains
    (9, 1, 3, MODE_REFERENCE),   # 81 slots, a candidate per cell
    (6, 3, 20, MODE_DENSE),      # 108 slots, three predictors of a cell in one class
    (5, 3, 7, MODE_DENSE),       # 75 slots
    (7, 2, 80, MODE_DENSE),      # a compile-time shape with more than 64 classes, logits in registers
])
@pytest.mark.parametrize("ct,it", [(0.0, 0.5), (0.3, 0.05), (0.5, 0.7)])
def test_two_candidates_per_lane(dev, S, Bx, C, mode, ct, it):
    """65 to 128 slots per image: one wavefront holds two candidates per lane and 128-bit
    lane masks in two words.  Random heads, heads whose boxes crowd (many IoUs above the
    threshold) and heads with equal scores, channel-major and cell-major, against the oracle."""
    head = YOLOv1Head(img_size=32 * S, n_cells=S, n_bboxes=Bx, n_classes=C)
    for variant in range(3):
        pred = synth.make_pred(9, S, Bx, C, seed=1000 * S + 10 * Bx + variant)
        if variant == 1:      # wide boxes: most pairs of a class overlap
            pred[:, [5 * k + q for k in range(Bx) for q in (2, 3)]] += 2.0
        if variant == 2:      # equal confidences: the order falls back to the flat index
            pred[:, [5 * k + 4 for k in range(Bx)]] = pred[:, [5 * k + 4 for k in range(Bx)]].round()
        pred = pred.to(dev)
        ltrb, score, cls, present = head.decode_candidates(pred, mode)
        want = oracle_nms(ltrb, score, cls, present, ct, it)
        assert_same(head.postprocess(pred, ct, it, mode, return_idx=True), want)
        nhwc = pred.contiguous(memory_format=torch.channels_last)
        assert_same(head.postprocess(nhwc, ct, it, mode, return_idx=True), want)


@pytest.mark.parametrize("S,Bx,C,mode", [
    (7, 2, 80, MODE_REFERENCE),   # 49 slots, 80 classes: two classes per lane in the class scan
    (8, 1, 128, MODE_DENSE),      # 64 slots, the largest class table without a sort
    (8, 1, 129, MODE_DENSE),      # one class more: the ranked path
    (7, 2, 65, MODE_REFERENCE),
    (5, 2, 100, MODE_DENSE),      # 50 slots
])
@pytest.mark.parametrize("ct,it", [(0.0, 0.5), (0.3, 0.05), (0.5, 0.7)])
def test_one_candidate_per_lane_with_more_than_64_classes(dev, S, Bx, C, mode, ct, it):
    """At most 64 slots and 65 to 128 classes: the class -> lane-mask table no longer fits the
    key array and the per-class row offsets are scanned two classes per lane."""
    head = YOLOv1Head(img_size=32 * S, n_cells=S, n_bboxes=Bx, n_classes=C)
    for variant in range(3):
        pred = synth.make_pred(9, S, Bx, C, seed=2000 * S + 10 * Bx + variant)
        if variant == 1:      # the upper classes win most cells, boxes crowd
            pred[:, 5 * Bx + C - 3:] += 3.0
            pred[:, [5 * k + q for k in range(Bx) for q in (2, 3)]] += 2.0
        if variant == 2:      # equal confidences
            pred[:, [5 * k + 4 for k in range(Bx)]] = pred[:, [5 * k + 4 for k in range(Bx)]].round()
        pred = pred.to(dev)
        ltrb, score, cls, present = head.decode_candidates(pred, mode)
        want = oracle_nms(ltrb, score, cls, present, ct, it)
        assert_same(head.postprocess(pred, ct, it, mode, return_idx=True), want)
        nhwc = pred.contiguous(memory_format=torch.channels_last)
        assert_same(head.postprocess(nhwc, ct, it, mode, return_idx=True), want)


def test_config1_against_full_cpu_path(dev, golden):
    """Whole inference path, CPU decode included, at config 1.  On this seed no
    tie case occurs, so the keep sets agree exactly; box values agree to the
    decode tolerance."""
    head = YOLOv1Head()
    pred = synth.make_pred(16, seed=0)
    g = golden("g7_oracle_cfg1.npz")
    for mode, ct, it, tag in ((MODE_REFERENCE, 0.5, 0.5, "mode0"), (MODE_DENSE, 0.1, 0.3, "mode1")):
        dets, count, kidx = head.postprocess(pred.to(dev), ct, it, mode, return_idx=True)
        np.testing.assert_array_equal(count.cpu().numpy(), g[f"count_{tag}"])
        np.testing.assert_array_equal(kidx.cpu().numpy(), g[f"keep_idx_{tag}"])
        np.testing.assert_allclose(dets.cpu().numpy(), g[f"dets_{tag}"], rtol=1e-5, atol=2e-4)


def test_full_size_batch_4096(dev):
    head = YOLOv1Head()
    pred = synth.make_pred(4096, seed=0).to(dev)
    ct, it = 0.5, 0.5
    dets, count, kidx = head.postprocess(pred, ct, it, MODE_REFERENCE, return_idx=True)
    dets2, count2, kidx2 = head.postprocess(pred, ct, it, MODE_REFERENCE, return_idx=True)
    assert torch.equal(dets, dets2) and torch.equal(count, count2) and torch.equal(kidx, kidx2)
    ltrb, score, cls, present = head.decode_candidates(pred, MODE_REFERENCE)
    # a 256-image slice against the oracle, bit-exact
    sl = slice(1000, 1256)
    assert_same((dets[sl], count[sl], kidx[sl]),
                oracle_nms(ltrb[sl], score[sl], cls[sl], present[sl], ct, it))
    # size-independent properties over the whole batch
    n = dets.shape[1]
    row = torch.arange(n, device=dev)[None, :]
    live = row < count[:, None]
    assert torch.equal(live, kidx >= 0)
    assert (dets[~live] == 0).all()
    assert (dets[..., 4][live] >= ct).all()
    # order: class ascending, then score descending, then flat index ascending
    c, s = dets[..., 5], dets[..., 4]
    pair = live[:, 1:] & live[:, :-1]
    ok = (c[:, 1:] > c[:, :-1]) | ((c[:, 1:] == c[:, :-1]) & (
        (s[:, 1:] < s[:, :-1]) | ((s[:, 1:] == s[:, :-1]) & (kidx[:, 1:] > kidx[:, :-1]))))
    assert ok[pair].all()
    # kept rows are the candidates they claim to be
    k = kidx.clamp(min=0).long()
    assert torch.equal(torch.gather(score, 1, k)[live], s[live])
    assert torch.equal(torch.gather(ltrb, 1, k[..., None].expand(-1, -1, 4))[live], dets[..., :4][live])
    # no kept pair of one class overlaps above the threshold
    b = dets[..., :4]
    iou = ref_cpu.get_iou  # pairwise restatement, evaluated on the GPU tensors per image chunk
    worst = 0.0
    for lo in range(0, 4096, 512):
        bb = b[lo:lo + 512]
        a1 = ref_cpu.get_area(bb)
        l = torch.maximum(bb[:, :, None, 0], bb[:, None, :, 0]); t = torch.maximum(bb[:, :, None, 1], bb[:, None, :, 1])
        r = torch.minimum(bb[:, :, None, 2], bb[:, None, :, 2]); bt = torch.minimum(bb[:, :, None, 3], bb[:, None, :, 3])
        inter = (r - l).clamp(min=0) * (bt - t).clamp(min=0)
        union = a1[:, :, None] + a1[:, None, :] - inter
        m = torch.where(union == 0, torch.zeros_like(union), inter / union)
        same = (c[lo:lo + 512, :, None] == c[lo:lo + 512, None, :]) & live[lo:lo + 512, :, None] & live[lo:lo + 512, None, :]
        same &= ~torch.eye(n, dtype=torch.bool, device=dev)[None]
        worst = max(worst, float(m[same].max()) if same.any() else 0.0)
    assert worst <= it
    # idempotence: NMS of the kept rows keeps all of them, in place
    d3, c3, k3 = YOLOv1Head.nms(dets[..., :4].contiguous(), dets[..., 4].contiguous(),
                                dets[..., 5].to(torch.int32), live, ct, it, return_idx=True)
    assert torch.equal(c3, count) and torch.equal(d3, dets)


def test_detections_list_and_empty_batch(dev):
    head = YOLOv1Head()
    pred = synth.make_pred(3, seed=4).to(dev)
    lst = head.detections(pred, 0.5, 0.5)
    dets, count = head.postprocess(pred, 0.5, 0.5)
    assert [x.shape[0] for x in lst] == count.cpu().tolist()
    assert all(x.shape[1] == 6 for x in lst)
    d0, c0 = head.postprocess(torch.zeros(0, 30, 7, 7, device=dev))
    assert d0.shape == (0, 98, 6) and c0.numel() == 0


@pytest.mark.parametrize("shape", [(7, 2, 20), (6, 3, 80), (14, 2, 20), (14, 5, 80)])
def test_row_count_chosen_by_the_caller(dev, shape):
    """yh_decode_nms_rows: reference mode keeps one box per cell (model.py:384-390), so S*S
    rows per image hold every detection; the rows, counts and indices equal those of
    yh_decode_nms, only the zero tail is shorter."""
    from yolo_amd import _lib
    S, Bx, C = shape
    head = YOLOv1Head(img_size=32 * S, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_pred(9, S, Bx, C, seed=21).to(dev)
    for mode, conf in ((MODE_REFERENCE, 0.5), (MODE_REFERENCE, 0.0), (MODE_DENSE, 0.3)):
        full = head.postprocess(pred, conf, 0.5, mode, return_idx=True)
        nslot = S * S if mode == MODE_REFERENCE else S * S * Bx
        for rows in (nslot, nslot + 3):
            dets, count, kidx = head.postprocess(pred, conf, 0.5, mode, return_idx=True, rows=rows)
            assert dets.shape == (9, rows, 6) and kidx.shape == (9, rows)
            n = min(rows, full[0].shape[1])
            assert torch.equal(count, full[1])
            assert torch.equal(dets[:, :n].view(torch.int32), full[0][:, :n].view(torch.int32))
            assert torch.equal(kidx[:, :n], full[2][:, :n])
            assert (dets[:, n:] == 0).all() and (kidx[:, n:] == -1).all()
            assert int(count.max()) <= nslot
        with pytest.raises(_lib.YoloHeadError) as e:
            head.postprocess(pred, conf, 0.5, mode, rows=nslot - 1)
        assert e.value.rc == _lib.YH_ERR_ARG
