"""Loss and decode+NMS kernels over ONE resident buffer set (what bench.py does)
against a rotation over many sets whose total size exceeds the 256 MB memory-side
cache, so that every launch reads and writes HBM proper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import synth, _lib
dev = torch.device("cuda:0")
B = 4096
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
base = [t.to(dev) for t in synth.make_inputs(B)]
def make_set():
    pred, gx, gc, om = (t.clone() for t in base)
    return dict(pred=pred, gx=gx, gc=gc, om=om.view(torch.uint8), grad=torch.empty_like(pred),
                dets=torch.empty(B, 98, 6, device=dev), cnt=torch.empty(B, dtype=torch.int32, device=dev))
ws = torch.zeros(lib.yh_workspace_bytes(0, B, 7, 2, 20), dtype=torch.uint8, device=dev)
def loss(s): lib.yh_loss_partials(s["pred"].data_ptr(), s["gx"].data_ptr(), s["gc"].data_ptr(), s["om"].data_ptr(), s["grad"].data_ptr(), ws.data_ptr(), ws.numel(), B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st)
def nms(s): lib.yh_decode_nms(s["pred"].data_ptr(), s["dets"].data_ptr(), s["cnt"].data_ptr(), None, 0, 0.5, 0.5, B, 7, 2, 20, 448, st)
def b2b(fn, sets, iters=320):
    for i in range(64): fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(sets[i % len(sets)])
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
for nsets in (1, 2, 4, 8, 16):
    sets = [make_set() for _ in range(nsets)]
    mb = nsets * sum(v.numel() * v.element_size() for v in sets[0].values()) / 1e6
    print(f"{nsets:2d} buffer sets ({mb:6.0f} MB): loss {b2b(loss, sets):6.2f} us  decode+NMS {b2b(nms, sets):6.2f} us")
    del sets
