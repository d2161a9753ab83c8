"""How much of a step is launch overhead: eager launches, a 1-step HIP graph and
an N-step HIP graph of the same three kernels, timed by wall clock."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, MODE_REFERENCE, LossLog
dev = torch.device("cuda:0")
head = YOLOv1Head()
pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(4096))
log = LossLog(dev, 64)
def step():
    t, g = head.loss_fwd_bwd(pred, gx, gc, om, log=log)
    d, c = head.postprocess(pred, 0.5, 0.5, MODE_REFERENCE)
    return t, g, d, c
def wall(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): step()
torch.cuda.synchronize()
print("eager          host/step %.1f us  total/step %.1f us" % wall(step, 300))
for nsteps in (1, 4, 16, 64):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(nsteps): outs = step()
    for _ in range(3): g.replay()
    h, t = wall(g.replay, max(4, 320 // nsteps))
    print("graph x%-3d     host/replay %.1f us  total/step %.2f us" % (nsteps, h, t / nsteps))
