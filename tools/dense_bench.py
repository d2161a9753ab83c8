"""decode+NMS with 65 to 128 slots per image (two candidates per lane of the one-wavefront
kernel), back-to-back HIP-event timing at batch 4096: shapes x modes x confidence thresholds.

    [YOLO_AMD_LIB=build/other.so] python tools/dense_bench.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import synth, _lib
dev = torch.device("cuda:0")
B = 4096
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
def b2b(fn, iters=300):
    for _ in range(100): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
out = []
for S, Bx, C, mode in ((7, 2, 20, 1), (8, 2, 20, 1), (7, 2, 24, 1), (10, 1, 20, 0), (6, 3, 80, 1), (7, 2, 80, 0), (8, 1, 128, 1), (7, 2, 24, 0), (7, 2, 80, 1)):
    pred = synth.make_pred(B, S, Bx, C, seed=3).to(dev)
    dets = torch.empty(B, S * S * Bx, 6, device=dev); cnt = torch.empty(B, dtype=torch.int32, device=dev)
    for conf in (0.5, 0.0):
        def k(): 
            rc = lib.yh_decode_nms(pred.data_ptr(), dets.data_ptr(), cnt.data_ptr(), None, mode, conf, 0.5, B, S, Bx, C, 32 * S, st)
            assert rc == 0, rc
        us = b2b(k)
        out.append(f"S={S},Bx={Bx},C={C} mode {mode} conf {conf}: {us:.2f} us (kept/image {cnt.float().mean().item():.1f})")
print(os.environ.get("YOLO_AMD_LIB", "in-tree"), *out, sep="\n  ")
