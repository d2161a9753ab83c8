"""Back-to-back duration of a kernel that does nothing (yh_scale_grad with scale == 1
returns at once) and of the finalize kernel: the fixed cost inside every b2b number."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
g = torch.zeros(4096 * 1470, device=dev); one = torch.ones((), device=dev)
def k(): lib.yh_scale_grad(g.data_ptr(), one.data_ptr(), g.numel(), st)
def b2b(fn, iters=500):
    for _ in range(100): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
print("early-exit kernel, 2048 workgroups: %.2f us per launch" % b2b(k))
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(64): k()
print("same inside a HIP graph (64 per graph): %.2f us per launch" % (b2b(gr.replay, 50) / 64))
