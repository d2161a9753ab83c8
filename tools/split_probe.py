"""Does splitting the batch over two streams (so one half loads while the other
stores) beat one launch over the whole batch?  Loss kernel only, graphs of 16."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, MODE_REFERENCE
dev = torch.device("cuda:0")
B = 4096
pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(B))
N = 16
def make(nsplit, what):
    heads = [YOLOv1Head() for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    graphs = []
    step = B // nsplit
    for i in range(nsplit):
        sl = slice(i * step, (i + 1) * step)
        p, x, c, m = pred[sl], gx[sl], gc[sl], om[sl]
        def fn(h=heads[i], p=p, x=x, c=c, m=m):
            if what in ("loss", "both"): h.loss_fwd_bwd(p, x, c, m, global_batch=B)
            if what in ("nms", "both"): h.postprocess(p, 0.5, 0.5, MODE_REFERENCE)
        with torch.cuda.stream(streams[i]):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[i]):
                for _ in range(N): fn()
        graphs.append(g)
    def run():
        for s, g in zip(streams, graphs):
            with torch.cuda.stream(s): g.replay()
    return run
def timeit(fn, reps=200):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps / N
for what in ("loss", "nms", "both"):
    for ns in (1, 2, 4):
        print(f"{what:5s} split {ns}: {timeit(make(ns, what)):.2f} us/step")

# loss and nms on separate streams, each split into `ns` batch slices (2*ns streams in all)
def make_sep(ns):
    runs = [make(ns, "loss"), make(ns, "nms")]
    def run():
        for r in runs: r()
    return run
for ns in (1, 2, 4):
    print(f"loss||nms, {ns} slices each ({2 * ns} streams): {timeit(make_sep(ns)):.2f} us/step")
