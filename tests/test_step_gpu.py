"""yh_head_step / yh_head_step_pipelined: the loss forward+backward and decode+NMS of one
head tensor in one call (model.py:296-317 and :379-394 on the same forward output: the
validation loss of train.py:100-110 plus the detections an evaluation feeds to eval.get_map,
eval.py:145).  At S=7, Bx=2, C=20 in reference mode one
kernel reads the head tensor once for both; every output must be the BITS of the two
separate entry points, and so must the oracle's parity carry over unchanged."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import MODE_DENSE, MODE_REFERENCE, LossLog, YOLOv1Head, synth

pytestmark = pytest.mark.gpu


def separate(head, dev_inputs, conf, iou, mode, rows, gb=None):
    terms, grad = head.loss_fwd_bwd(*dev_inputs, global_batch=gb)
    dets, count, kidx = head.postprocess(dev_inputs[0], conf, iou, mode, return_idx=True, rows=rows)
    return terms, grad, dets, count, kidx


def same_bits(a, b):
    return a.shape == b.shape and torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


@pytest.mark.parametrize("batch", [1, 3, 4, 5, 16, 257, 4096])
def test_one_kernel_equals_the_two_launches(dev, batch):
    head = YOLOv1Head()
    d = [t.to(dev) for t in synth.make_inputs(batch, seed=30 + batch)]
    for conf, iou, rows in ((0.5, 0.5, 49), (0.0, 0.3, 49), (0.9, 0.5, None), (0.2, 0.0, 60)):
        want = separate(head, d, conf, iou, MODE_REFERENCE, rows)
        got = head.head_step(*d, conf, iou, MODE_REFERENCE, rows=rows, return_idx=True)
        for g, w in zip(got, want):
            assert same_bits(g, w)
    # a shard of a larger batch: the loss is divided by the global batch (model.py:317)
    want = separate(head, d, 0.5, 0.5, MODE_REFERENCE, 49, gb=3 * batch + 1)
    got = head.head_step(*d, 0.5, 0.5, MODE_REFERENCE, global_batch=3 * batch + 1, rows=49, return_idx=True)
    for g, w in zip(got, want):
        assert same_bits(g, w)


def test_against_the_oracle(dev):
    """The fused kernel directly against the CPU oracle (not only through the separate
    kernels): loss terms, gradient, kept rows."""
    head = YOLOv1Head()
    pred, gx, gc, om = synth.make_inputs(64, seed=5)
    terms, grad, dets, count, kidx = head.head_step(pred.to(dev), gx.to(dev), gc.to(dev), om.to(dev),
                                                    0.4, 0.5, rows=49, return_idx=True)
    ref = ref_cpu.loss_and_grad(pred, gx, gc, om)
    want = torch.stack([ref["coord"], ref["conf"], ref["cls"], ref["total"], ref["loss"]])
    torch.testing.assert_close(terms.cpu(), want, rtol=1e-5, atol=0)           # 1e-5 relative
    scale = ref["grad"].abs().max().item()
    assert (grad.cpu() - ref["grad"]).abs().max().item() <= 1e-5 * scale        # 1e-5 of the largest entry
    ltrb, score, cls, present = head.decode_candidates(pred.to(dev), MODE_REFERENCE)
    wd, wc, wk = ref_cpu.nms_batch(ltrb.cpu().numpy(), score.cpu().numpy(), cls.cpu().numpy(),
                                   present.cpu().numpy(), 0.4, 0.5)
    np.testing.assert_array_equal(count.cpu().numpy(), wc)                       # bit-exact keep sets
    np.testing.assert_array_equal(kidx.cpu().numpy(), wk[:, :49])
    np.testing.assert_array_equal(dets.cpu().numpy().view(np.uint32), wd[:, :49].view(np.uint32))


def test_many_objects_and_none(dev):
    """Images with no object, with more listed cells than the parked ground truth has slots
    for (33 to 49 per image), a crowded class and extreme logits, side by side."""
    head = YOLOv1Head()
    batch = 11
    g = torch.Generator().manual_seed(9)
    pred = synth.make_pred(batch, seed=4)
    pred[3] = pred[3] * 40            # saturated sigmoids
    pred[5, 10:] = 0.0
    pred[5, 12] = 5.0                 # one class everywhere: 49 members
    pred[5, [4, 9]] = 3.0
    gx = torch.rand(batch, 49, 1, 4, generator=g) * 0.9 + 0.05
    gc = torch.softmax(torch.randn(batch, 49, 1, 20, generator=g) * 3, dim=-1)
    om = torch.zeros(batch, 49, 1, 1, dtype=torch.bool)
    for b, n in enumerate([0, 1, 31, 32, 33, 40, 48, 49, 49, 2, 0]):
        om[b, torch.randperm(49, generator=g)[:n]] = True
    gx, gc = gx * om, gc * om
    d = [t.to(dev) for t in (pred, gx, gc, om)]
    for conf in (0.0, 0.5):
        want = separate(head, d, conf, 0.4, MODE_REFERENCE, 49)
        got = head.head_step(*d, conf, 0.4, MODE_REFERENCE, rows=49, return_idx=True)
        for gg, w in zip(got, want):
            assert same_bits(gg, w)
    assert torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all()
    ref = ref_cpu.loss_and_grad(pred, gx, gc, om)
    torch.testing.assert_close(got[0][4].cpu(), ref["loss"], rtol=1e-5, atol=0)


@pytest.mark.parametrize("shape,mode", [((7, 2, 20), MODE_DENSE), ((7, 3, 20), MODE_REFERENCE),
                                        ((14, 5, 80), MODE_DENSE), ((6, 2, 24), MODE_REFERENCE)])
def test_other_shapes_and_modes_take_the_two_launches(dev, shape, mode):
    S, Bx, C = shape
    head = YOLOv1Head(img_size=32 * S, n_cells=S, n_bboxes=Bx, n_classes=C)
    d = [t.to(dev) for t in synth.make_inputs(7, S, Bx, C, 32 * S, seed=8)]
    want = separate(head, d, 0.3, 0.5, mode, None)
    got = head.head_step(*d, 0.3, 0.5, mode, return_idx=True)
    for g, w in zip(got, want):
        assert same_bits(g, w)


def test_pipelined_step_logs_every_step_one_call_late(dev):
    """One launch per step; steps of different batches alternate with plain pipelined loss
    steps on the same workspace, and the log holds every step's sums in order, with the
    bits of the two-launch path."""
    head, ref_head = YOLOv1Head(), YOLOv1Head()
    log, ref_log = LossLog(dev, 16), LossLog(dev, 16)
    sizes = [4096, 7, 64, 4096, 1, 333]
    for i, b in enumerate(sizes):
        d = [t.to(dev) for t in synth.make_inputs(b, seed=60 + i)]
        want = separate(ref_head, d, 0.5, 0.5, MODE_REFERENCE, 49)
        ref_head.loss_fwd_bwd(*d, log=ref_log)
        if i % 3 == 2:
            grad = head.loss_step_pipelined(*d, log)
        else:
            grad, dets, count = head.head_step_pipelined(*d, log, 0.5, 0.5, rows=49)
            assert same_bits(dets, want[2]) and same_bits(count, want[3])
        assert same_bits(grad, want[1])
    head.flush_loss_log(log)
    torch.cuda.synchronize()
    assert int(log.cursor) == len(sizes)
    assert torch.equal(log.ring, ref_log.ring)


def test_pipelined_step_under_graph_replay(dev):
    """An even number of fused steps captured once and replayed: every replay appends its
    steps to the device-side log."""
    from yolo_amd import _lib
    lib = _lib.load()
    batch, S, Bx, C, img = 512, 7, 2, 20, 448
    pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(batch, seed=71))
    om8 = om.view(torch.uint8)
    head = YOLOv1Head()
    want = separate(head, [pred, gx, gc, om], 0.5, 0.5, MODE_REFERENCE, 49)
    grad, dets = torch.empty_like(pred), torch.empty(batch, 49, 6, device=dev)
    cnt = torch.empty(batch, dtype=torch.int32, device=dev)
    log = LossLog(dev, 32)
    ws = torch.zeros(lib.yh_workspace_bytes(_lib.YH_OP_LOSS_PIPELINED, batch, S, Bx, C), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(dev)

    def steps(n):
        for i in range(n):
            _lib.check(lib.yh_head_step_pipelined(
                pred.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(),
                dets.data_ptr(), cnt.data_ptr(), None, None, log.ring.data_ptr(), log.capacity,
                log.cursor.data_ptr(), ws.data_ptr(), ws.numel(), i & 1, 0, 0.5, 0.5, 49, batch, S, Bx, C,
                img, 5.0, 0.5, 1.0 / batch, side.cuda_stream))
        _lib.check(lib.yh_loss_pipelined_flush(ws.data_ptr(), ws.numel(), (n - 1) & 1, None, log.ring.data_ptr(),
                                               log.capacity, log.cursor.data_ptr(), side.cuda_stream))

    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        steps(2)
    torch.cuda.synchronize()
    log.reset()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        steps(4)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert int(log.cursor) == 12
    assert torch.equal(log.ring[:12], want[0][:4].expand(12, 4))
    assert same_bits(grad, want[1]) and same_bits(dets, want[2]) and same_bits(cnt, want[3])


def test_argument_errors(dev):
    from yolo_amd import _lib
    lib = _lib.load()
    f = 0x1000       # never dereferenced: the host-side checks fail first
    base = dict(batch=8, S=7, Bx=2, C=20, img=448)

    def call(rows=49, mode=0, conf=0.5, ws=f, ws_bytes=1 << 20, terms=f, dets=f, **kw):
        a = dict(base, **kw)
        return lib.yh_head_step(f, f, f, f, f, terms, dets, f, None, ws, ws_bytes, mode, conf, 0.5, rows,
                                a["batch"], a["S"], a["Bx"], a["C"], a["img"], 5.0, 0.5, 1.0, None)

    assert call(rows=48) == _lib.YH_ERR_ARG            # fewer rows than candidates
    assert call(rows=97, mode=1) == _lib.YH_ERR_ARG
    assert call(mode=2) == _lib.YH_ERR_ARG
    assert call(conf=float("nan")) == _lib.YH_ERR_ARG
    assert call(terms=None) == _lib.YH_ERR_NULL
    assert call(dets=None) == _lib.YH_ERR_NULL
    assert call(ws=None) == _lib.YH_ERR_WORKSPACE
    assert call(ws_bytes=8) == _lib.YH_ERR_WORKSPACE
    assert call(img=450) == _lib.YH_ERR_SHAPE
    assert call(S=64, Bx=16, img=64) == _lib.YH_ERR_UNSUPPORTED
