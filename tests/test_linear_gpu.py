"""The last layer's bias and view folded into the head read (model.py:133-135,
146-148).  Reference semantics: out = (x @ W.T + bias).view(-1, D, S, S); the fused
entry points get z = x @ W.T and bias separately.  Bar: identical bits to the
unfused entry points on z + bias (the kernel performs the same fp32 addition
first), bias gradient within 1e-6 relative of an fp64 column sum."""
import pytest
import torch

from oracle import ref_cpu
from yolo_amd import MODE_DENSE, MODE_REFERENCE, YOLOv1Head, _lib, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,Bx,C,img,batch", [(7, 2, 20, 448, 4096), (7, 2, 20, 448, 37),
                                              (14, 5, 80, 448, 5), (3, 1, 4, 96, 9)])
def test_loss_on_product_plus_bias(dev, S, Bx, C, img, batch):
    head = YOLOv1Head(img_size=img, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred, gx, gc, om = (x.to(dev) for x in synth.make_inputs(batch, S, Bx, C, img, seed=S + batch))
    D = 5 * Bx + C
    bias = torch.randn(D * S * S, generator=torch.Generator().manual_seed(3)).to(dev)
    z = pred.view(batch, -1) - bias                       # so that z + bias is a realistic head tensor
    out = (z + bias).view(batch, D, S, S)
    terms, grad_z, grad_bias = head.loss_fwd_bwd_linear(z, bias, gx, gc, om)
    want_terms, want_grad = head.loss_fwd_bwd(out, gx, gc, om)
    assert torch.equal(terms, want_terms)
    assert grad_z.shape == z.shape and torch.equal(grad_z.view_as(want_grad), want_grad)
    want_gb = want_grad.double().view(batch, -1).sum(0)
    assert (grad_bias.double() - want_gb).abs().max().item() <= 1e-6 * want_gb.abs().max().item()
    # fixed summation order: the same bits again
    _, _, again = head.loss_fwd_bwd_linear(z, bias, gx, gc, om)
    assert torch.equal(again, grad_bias)


def test_against_the_oracle_through_a_linear_layer(dev):
    """nn.Linear -> view -> reference loss (oracle, CPU autograd) against
    product + bias -> fused head, gradients of weight, bias and input included."""
    torch.manual_seed(0)
    head = YOLOv1Head()
    lin = torch.nn.Linear(64, 1470)
    x = torch.randn(16, 64)
    _, gx, gc, om = synth.make_inputs(16, seed=21)
    xc = x.clone().requires_grad_(True)
    ref_loss = ref_cpu.get_loss(lin(xc).view(-1, 30, 7, 7), gx, gc, om)
    ref_loss.backward()
    ref_gw, ref_gb = lin.weight.grad.clone(), lin.bias.grad.clone()
    lin_g = torch.nn.Linear(64, 1470).to(dev)
    lin_g.load_state_dict(lin.state_dict())
    xg = x.to(dev).requires_grad_(True)
    z = torch.nn.functional.linear(xg, lin_g.weight)            # no bias here ...
    loss = head.get_loss_linear(z, lin_g.bias, gx.to(dev), gc.to(dev), om.to(dev))   # ... it is added in the kernel
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
    for got, want in ((lin_g.weight.grad, ref_gw), (lin_g.bias.grad, ref_gb), (xg.grad, xc.grad)):
        assert (got.cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("S,Bx,C,batch,mode,ct", [(7, 2, 20, 257, MODE_REFERENCE, 0.5),
                                                  (7, 2, 20, 33, MODE_DENSE, 0.1),
                                                  (14, 5, 80, 3, MODE_DENSE, 0.001),
                                                  (11, 3, 7, 5, MODE_REFERENCE, 0.3)])
def test_decode_nms_on_product_plus_bias(dev, S, Bx, C, batch, mode, ct):
    head = YOLOv1Head(img_size=32 * S, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_pred(batch, S, Bx, C, seed=9 * S).to(dev)
    D = 5 * Bx + C
    bias = torch.randn(D * S * S, generator=torch.Generator().manual_seed(5)).to(dev)
    z = pred.view(batch, -1) - bias
    out = (z + bias).view(batch, D, S, S)
    got = head.postprocess(z.view(batch, D, S, S), ct, 0.5, mode, return_idx=True, bias=bias)
    want = head.postprocess(out, ct, 0.5, mode, return_idx=True)
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_bias_grad_and_argument_errors(dev):
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for nrows, ncols in ((0, 7), (1, 1), (5, 1470), (33, 100), (4097, 65), (70000, 3)):
        g = torch.randn(nrows, ncols, device=dev)
        out = torch.full((ncols,), float("nan"), device=dev)
        ws = torch.empty(max(lib.yh_bias_grad_workspace_bytes(nrows, ncols), 4), dtype=torch.uint8, device=dev)
        _lib.check(lib.yh_bias_grad(g.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), nrows, ncols, st))
        want = g.double().sum(0)
        tol = 1e-6 * max(float(g.abs().sum(0).max()) if nrows else 0.0, 1e-30)
        assert (out.double() - want).abs().max().item() <= tol
    f = out.data_ptr()
    assert lib.yh_bias_grad(f, f, None, 0, 4, 4, st) == _lib.YH_ERR_WORKSPACE
    assert lib.yh_bias_grad(f, None, f, 1 << 20, 4, 4, st) == _lib.YH_ERR_NULL
    assert lib.yh_bias_grad(f, f, f, 1 << 20, 4, 0, st) == _lib.YH_ERR_SHAPE
    assert lib.yh_loss_fwd_bwd_linear(f, None, f, f, f, f, f, f, 1 << 20, 4, 7, 2, 20, 448, 5.0, 0.5,
                                      0.25, st) == _lib.YH_ERR_NULL
    assert lib.yh_decode_nms_linear(f, None, f, f, None, 0, 0.5, 0.5, 4, 7, 2, 20, 448, st) == _lib.YH_ERR_NULL
    torch.cuda.synchronize()
