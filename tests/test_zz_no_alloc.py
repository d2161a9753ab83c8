"""yh_loss_fwd_bwd_static (loss rows in a static device array, no allocation, no
workspace argument) against yh_loss_fwd_bwd through the C ABI: same kernels, so the
five loss terms and the gradient must be the same bits.  Two calls in a row, the
second on a smaller input: the array keeps the first call's rows, and the second
must not read them.

The first call on a device resolves the array's address on the host (hipGetDevice,
hipGetSymbolAddress); it must happen outside graph capture.  Every later call only
enqueues kernels and can be captured."""
import pytest
import torch

from yolo_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _both(lib, dev, batch, S, Bx, C, img, seed):
    pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(batch, S, Bx, C, img, seed=seed))
    om8 = om.view(torch.uint8)
    st = torch.cuda.current_stream(dev).cuda_stream
    ws = torch.empty(max(16, lib.yh_workspace_bytes(_lib.YH_OP_LOSS, batch, S, Bx, C)),
                     dtype=torch.uint8, device=dev)
    grad_ref = torch.full_like(pred, float("nan"))
    terms_ref = torch.full((5,), float("nan"), device=dev)
    _lib.check(lib.yh_loss_fwd_bwd(pred.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(),
                                   grad_ref.data_ptr(), terms_ref.data_ptr(), ws.data_ptr(),
                                   ws.numel(), batch, S, Bx, C, img, 5.0, 0.5, 1.0 / batch, st))
    grad = torch.full_like(pred, float("nan"))
    terms = torch.full((5,), float("nan"), device=dev)
    _lib.check(lib.yh_loss_fwd_bwd_static(pred.data_ptr(), gx.data_ptr(), gc.data_ptr(),
                                          om8.data_ptr(), grad.data_ptr(), terms.data_ptr(),
                                          batch, S, Bx, C, img, 5.0, 0.5, 1.0 / batch, st))
    torch.cuda.synchronize()
    assert torch.isfinite(terms_ref).all() and torch.isfinite(grad_ref).all()
    assert torch.equal(terms, terms_ref), (terms, terms_ref)
    assert torch.equal(grad, grad_ref)


def test_static_scratch_matches_workspace_entry_point(dev):
    lib = _lib.load()
    _both(lib, dev, 4096, 7, 2, 20, 448, seed=5)      # 1,024 rows (four images per workgroup)
    _both(lib, dev, 33, 7, 2, 20, 448, seed=6)        # 9 rows: 1,015 stale rows stay behind
    _both(lib, dev, 64, 14, 5, 80, 448, seed=7)       # run-time-shape kernel, 64 rows
    _both(lib, dev, 3, 17, 2, 5, 34 * 17, seed=8)     # smaller again


def test_static_scratch_limits(dev):
    lib = _lib.load()
    f = 0x1000    # never dereferenced: the host-side checks fail first
    # S = 7: four images per row of 16 bytes, 16,384 rows fit: 65,536 images, not one more
    assert lib.yh_loss_fwd_bwd_static(f, f, f, f, f, f, 65537, 7, 2, 20, 448, 5.0, 0.5, 1.0,
                                      None) == _lib.YH_ERR_SCRATCH
    assert lib.yh_loss_fwd_bwd_static(f, f, f, f, f, f, 16385, 16, 2, 20, 448, 5.0, 0.5, 1.0,
                                      None) == _lib.YH_ERR_SCRATCH
    assert lib.yh_loss_fwd_bwd_static(f, f, f, f, f, None, 16, 7, 2, 20, 448, 5.0, 0.5, 1.0,
                                      None) == _lib.YH_ERR_NULL
    assert lib.yh_loss_fwd_bwd_static(f, f, f, f, f, f, 16, 7, 2, 20, 450, 5.0, 0.5, 1.0,
                                      None) == _lib.YH_ERR_SHAPE
    assert "static scratch" in _lib.strerror(_lib.YH_ERR_SCRATCH)


def test_static_scratch_at_its_limit(dev):
    """65,536 images at S = 7 fill the array to its last row."""
    _both(_lib.load(), dev, 65536, 7, 2, 20, 448, seed=9)


def test_static_scratch_under_graph_capture(dev):
    """After one eager call on the device (which resolves the array's address) the entry
    point only enqueues kernels: captured into a HIP graph and replayed it gives the bits
    of yh_loss_fwd_bwd."""
    lib = _lib.load()
    batch, S, Bx, C, img = 257, 7, 2, 20, 448
    pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(batch, S, Bx, C, img, seed=10))
    om8 = om.view(torch.uint8)
    ws = torch.empty(lib.yh_workspace_bytes(_lib.YH_OP_LOSS, batch, S, Bx, C), dtype=torch.uint8, device=dev)
    grad_ref, terms_ref = torch.empty_like(pred), torch.empty(5, device=dev)
    grad, terms = torch.empty_like(pred), torch.empty(5, device=dev)

    def call(stream):
        _lib.check(lib.yh_loss_fwd_bwd_static(pred.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(),
                                              grad.data_ptr(), terms.data_ptr(), batch, S, Bx, C, img,
                                              5.0, 0.5, 1.0 / batch, stream.cuda_stream))

    _lib.check(lib.yh_loss_fwd_bwd(pred.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(),
                                   grad_ref.data_ptr(), terms_ref.data_ptr(), ws.data_ptr(), ws.numel(),
                                   batch, S, Bx, C, img, 5.0, 0.5, 1.0 / batch,
                                   torch.cuda.current_stream(dev).cuda_stream))
    call(torch.cuda.current_stream(dev))          # eager: the address is resolved here
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        call(side)
    for _ in range(2):
        grad.fill_(float("nan"))
        terms.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(terms, terms_ref) and torch.equal(grad, grad_ref)
