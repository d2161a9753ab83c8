"""Cell-major (channels-last) head tensor against the channel-major one: loss and
decode+NMS through the C ABI, back-to-back HIP-event microseconds per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import synth, _lib
dev = torch.device("cuda:0")
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
def b2b(fn, iters=100):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
for S, BX, C, B in ((7, 2, 20, 4096), (7, 2, 24, 4096), (14, 2, 20, 1024), (14, 5, 80, 1024), (13, 5, 20, 1024)):
    pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(B, S, BX, C, 32 * S))
    om8 = om.view(torch.uint8)
    pcl = pred.permute(0, 2, 3, 1).contiguous()          # (B, S, S, D)
    ws = torch.zeros(lib.yh_workspace_bytes(0, B, S, BX, C), dtype=torch.uint8, device=dev)
    grad = torch.empty_like(pred); terms = torch.empty(5, device=dev)
    dets = torch.empty(B, S * S * BX, 6, device=dev); cnt = torch.empty(B, dtype=torch.int32, device=dev)
    def loss(fn, p): _lib.check(fn(p.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(), terms.data_ptr(), ws.data_ptr(), ws.numel(), B, S, BX, C, 32 * S, 5.0, 0.5, 1.0 / B, st))
    def nms(fn, p, mode=0, conf=0.5): _lib.check(fn(p.data_ptr(), dets.data_ptr(), cnt.data_ptr(), None, mode, conf, 0.5, B, S, BX, C, 32 * S, st))
    print(f"S={S:2d} B={BX} C={C:2d} batch {B}: loss {b2b(lambda: loss(lib.yh_loss_fwd_bwd, pred)):6.1f} us channel-major, "
          f"{b2b(lambda: loss(lib.yh_loss_fwd_bwd_nhwc, pcl)):6.1f} us cell-major | decode+NMS "
          f"{b2b(lambda: nms(lib.yh_decode_nms, pred)):6.1f} us channel-major, {b2b(lambda: nms(lib.yh_decode_nms_nhwc, pcl)):6.1f} us cell-major")
    print(f"      dense candidates, conf 0.001: decode+NMS {b2b(lambda: nms(lib.yh_decode_nms, pred, 1, 0.001)):6.1f} us channel-major, "
          f"{b2b(lambda: nms(lib.yh_decode_nms_nhwc, pcl, 1, 0.001)):6.1f} us cell-major")
