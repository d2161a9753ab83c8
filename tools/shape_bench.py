"""Loss and decode+NMS timings for shapes other than the benchmark one (they run
the run-time-shape kernels): back-to-back HIP-event microseconds per call."""
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
for S, BX, C, B in ((7, 2, 20, 4096), (7, 2, 24, 4096), (7, 3, 20, 4096), (7, 2, 80, 4096), (14, 2, 20, 1024), (14, 5, 80, 1024), (13, 5, 20, 1024)):
    pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(B, S, BX, C, 32 * S))
    om8 = om.view(torch.uint8)
    D = 5 * BX + C
    lb = B * (2 * 4 * D * S * S + S * S * (4 * (4 + C) + 1))
    nb_ = B * (4 * D * S * S + 24 * S * S * BX + 4)
    # straight through the C ABI on preallocated buffers: device time, no allocator in the loop
    ws = torch.zeros(lib.yh_workspace_bytes(0, B, S, BX, C), dtype=torch.uint8, device=dev)
    grad = torch.empty_like(pred); terms = torch.empty(5, device=dev)
    dets = torch.empty(B, S * S * BX, 6, device=dev); cnt = torch.empty(B, dtype=torch.int32, device=dev)
    def loss(): _lib.check(lib.yh_loss_fwd_bwd(pred.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(), terms.data_ptr(), ws.data_ptr(), ws.numel(), B, S, BX, C, 32 * S, 5.0, 0.5, 1.0 / B, st))
    def nms(): _lib.check(lib.yh_decode_nms(pred.data_ptr(), dets.data_ptr(), cnt.data_ptr(), None, 0, 0.5, 0.5, B, S, BX, C, 32 * S, st))
    tl = b2b(loss)
    tn = b2b(nms)
    print(f"S={S:2d} B={BX} C={C:2d} batch {B}: loss {tl:7.1f} us {lb / tl / 1e3:7.0f} GB/s | decode+NMS {tn:7.1f} us {nb_ / tn / 1e3:7.0f} GB/s")
