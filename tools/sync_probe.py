"""Where the fixed cost of a short timed region goes: a 20-step graph replay timed by
wall clock, with the default wait policy and with hipDeviceScheduleSpin."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 1 and sys.argv[1] == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
from yolo_amd import LossLog, synth
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
K = bench.Kernels(dev, 4096, 4096, True)
res = K.make_set(synth.make_inputs(4096))
log = LossLog(dev, 64)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def enqueue(n):
    s2.wait_stream(s1)
    for _ in range(n): K.loss(res, log, s1)
    K.loss_flush(log, s1)
    for _ in range(n): K.nms(res, s2)
    s1.wait_stream(s2)
with torch.cuda.stream(s1): enqueue(4)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s1): enqueue(20)
for _ in range(50):
    with torch.cuda.stream(s1): g.replay()
torch.cuda.synchronize()
out = []
for _ in range(20):
    torch.cuda.synchronize(); time.sleep(0.002)
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out.append((1e6 * (t1 - t0), 1e6 * (t2 - t0)))
out.sort(key=lambda x: x[1])
print("replay call returns after %.1f us, synchronized after %.1f us (median of 20; best %.1f)" % (out[10][0], out[10][1], out[0][1]))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s1):
    a.record(); g.replay(); b.record()
torch.cuda.synchronize()
print("GPU time of the 20-step graph by events: %.1f us" % (1e3 * a.elapsed_time(b)))
# the same 20 steps as two single-stream graphs, one launch per stream
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s1):
    for _ in range(20): K.loss(res, log, s1)
    K.loss_flush(log, s1)
with torch.cuda.graph(g2, stream=s2):
    for _ in range(20): K.nms(res, s2)
def two():
    with torch.cuda.stream(s1): g1.replay()
    with torch.cuda.stream(s2): g2.replay()
for _ in range(50): two()
torch.cuda.synchronize()
out = []
for _ in range(20):
    torch.cuda.synchronize(); time.sleep(0.002)
    t0 = time.perf_counter(); two(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out.append((1e6 * (t1 - t0), 1e6 * (t2 - t0)))
out.sort(key=lambda x: x[1])
print("two graphs: replay calls return after %.1f us, synchronized after %.1f us (median of 20; best %.1f)" % (out[10][0], out[10][1], out[0][1]))
