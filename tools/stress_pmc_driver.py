"""A few launches of the stress-shape decode+NMS kernel (S=14, Bx=5, C=80, conf 0.001, dense
candidates, batch 1024) and of the run-time-shape loss kernel at that shape, straight
through the C ABI: the program the rocprofv3 --pmc passes for that shape run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import synth, _lib
dev = torch.device("cuda:0")
B, S, BX, C, IMG = 1024, 14, 5, 80, 448
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(B, S, BX, C, IMG))
om8 = om.view(torch.uint8)
dets = torch.empty(B, S * S * BX, 6, device=dev); cnt = torch.empty(B, dtype=torch.int32, device=dev)
ws = torch.zeros(lib.yh_workspace_bytes(0, B, S, BX, C), dtype=torch.uint8, device=dev)
grad = torch.empty_like(pred); terms = torch.empty(5, device=dev)
for _ in range(12):
    _lib.check(lib.yh_decode_nms(pred.data_ptr(), dets.data_ptr(), cnt.data_ptr(), None, 1, 0.001, 0.5, B, S, BX, C, IMG, st))
    _lib.check(lib.yh_loss_fwd_bwd(pred.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(), terms.data_ptr(), ws.data_ptr(), ws.numel(), B, S, BX, C, IMG, 5.0, 0.5, 1.0 / B, st))
torch.cuda.synchronize()
print("kept per image", cnt.float().mean().item())
