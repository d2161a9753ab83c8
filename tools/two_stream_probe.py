"""Loss and decode+NMS are independent readers of the head tensor: time them
serialised on one stream and concurrently on two streams (16 steps per graph)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, MODE_REFERENCE, LossLog
dev = torch.device("cuda:0")
head = YOLOv1Head()
pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(4096))
log = LossLog(dev, 64)
N = 16
def f_loss(): return head.loss_fwd_bwd(pred, gx, gc, om, log=log)
def f_nms(): return head.postprocess(pred, 0.5, 0.5, MODE_REFERENCE)
def capture(fns, stream):
    with torch.cuda.stream(stream):
        for f in fns: f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(N):
                for f in fns: keep = f()
    return g
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g_both = capture([f_loss, f_nms], s1)
g_loss = capture([f_loss], s1)
g_nms = capture([f_nms], s2)
def run(fn, reps=200):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps / N
def serial():
    with torch.cuda.stream(s1): g_both.replay()
def concurrent():
    with torch.cuda.stream(s1): g_loss.replay()
    with torch.cuda.stream(s2): g_nms.replay()
def only_loss():
    with torch.cuda.stream(s1): g_loss.replay()
def only_nms():
    with torch.cuda.stream(s2): g_nms.replay()
for name, fn in (("serial one stream", serial), ("two streams", concurrent), ("loss only", only_loss), ("nms only", only_nms), ("serial one stream", serial), ("two streams", concurrent)):
    print("%-20s %.2f us/step" % (name, run(fn)))
