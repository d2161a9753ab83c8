"""The last layer's bias folded into the head read against adding it first
(batch 4096, back-to-back HIP-event timing, microseconds per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, _lib
dev = torch.device("cuda:0")
B = 4096
head = YOLOv1Head()
pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(B))
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
bias = torch.randn(1470, device=dev); z = pred.view(B, -1) - bias
om8 = om.view(torch.uint8)
ws = torch.zeros(max(lib.yh_workspace_bytes(0, B, 7, 2, 20), lib.yh_bias_grad_workspace_bytes(B, 1470)), dtype=torch.uint8, device=dev)
grad = torch.empty_like(pred); terms = torch.empty(5, device=dev); gb = torch.empty(1470, device=dev)
dets = torch.empty(B, 98, 6, device=dev); cnt = torch.empty(B, dtype=torch.int32, device=dev)
out = torch.empty_like(z)
def loss_lin(): lib.yh_loss_fwd_bwd_linear(z.data_ptr(), bias.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(), terms.data_ptr(), ws.data_ptr(), ws.numel(), B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st)
def loss_plain(): lib.yh_loss_fwd_bwd(out.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(), terms.data_ptr(), ws.data_ptr(), ws.numel(), B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st)
def add_then_loss(): torch.add(z, bias, out=out); loss_plain()
def nms_lin(): lib.yh_decode_nms_linear(z.data_ptr(), bias.data_ptr(), dets.data_ptr(), cnt.data_ptr(), None, 0, 0.5, 0.5, B, 7, 2, 20, 448, st)
def nms_plain(): lib.yh_decode_nms(out.data_ptr(), dets.data_ptr(), cnt.data_ptr(), None, 0, 0.5, 0.5, B, 7, 2, 20, 448, st)
def add_then_nms(): torch.add(z, bias, out=out); nms_plain()
def bias_grad(): lib.yh_bias_grad(grad.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws.numel(), B, 1470, st)
g2 = grad.view(B, 1470)
def torch_sum(): torch.sum(g2, 0, out=gb)
def b2b(fn, iters=300):
    for _ in range(100): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
for name, fn in (("loss, bias folded in", loss_lin), ("torch add + loss", add_then_loss), ("loss alone (two launches)", loss_plain),
                 ("decode+NMS, bias folded in", nms_lin), ("torch add + decode+NMS", add_then_nms), ("decode+NMS alone", nms_plain),
                 ("yh_bias_grad", bias_grad), ("torch.sum(grad, 0)", torch_sum)):
    print(f"{name:32s} {b2b(fn):7.2f} us")
