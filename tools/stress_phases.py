"""Phase timings of the stress-shape decode+NMS kernel from time stamps that thread 0
of every workgroup takes at the phase boundaries (probe build: make -C yolo_amd/csrc
probe; YOLO_AMD_LIB=build/libyolohead_probe.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, MODE_DENSE
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_BATCH", "1024"))
S, BX, C, IMG = 14, 5, 80, 448
head = YOLOv1Head(IMG, S, BX, C)
pred = synth.make_inputs(B, S, BX, C, IMG)[0].to(dev)
CROWD = float(os.environ.get("KB_CROWD", "0"))      # one class wins this share of the cells
if CROWD > 0:
    mask = torch.rand(B, S, S, device=dev, generator=torch.Generator(dev).manual_seed(1)) < CROWD
    pred[:, 5 * BX + 40][mask] += 30.0
stamps = torch.zeros(B, 16, dtype=torch.int64, device=dev)
for _ in range(5): head.postprocess(pred, 0.001, 0.5, MODE_DENSE)
os.environ["YH_NMS_STAMPS"] = str(stamps.data_ptr())
acc = torch.zeros(16, dtype=torch.float64)
first = torch.zeros(16, dtype=torch.float64)
N = 20
for _ in range(N):
    stamps.zero_()
    head.postprocess(pred, 0.001, 0.5, MODE_DENSE)
    torch.cuda.synchronize()
    s = stamps.cpu().double()
    t0 = s[:, 0].min()
    acc += (s - t0).mean(0) / 100.0      # 100 MHz counter -> us
names = ["entry", "class argmax", "decode+keys", "hist", "scan a", "scan b", "bucket scatter", "rank",
         "C scatter", "numbering a", "numbering b", "pairs", "head list", "greedy walk", "alive", "output"]
acc /= N
prev = acc[0].item()
print("variant", os.environ.get("YH_NMS_VARIANT", "0"), "mean start offset %.2f us" % prev)
for i in range(1, 16):
    print("  %-16s +%.2f us  (at %.2f)" % (names[i], acc[i].item() - prev, acc[i].item()))
    prev = acc[i].item()
# spread over the workgroups of the last launch: when each one starts, leaves the memory phase and ends
s = (stamps.cpu().double() - stamps.cpu().double()[:, 0].min()) / 100.0
for name, col in (("start", 0), ("memory phase done", 2), ("pairs done", 11), ("end", 15)):
    q = torch.quantile(s[:, col], torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))
    print("  %-18s min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us" % ((name,) + tuple(q.tolist())))
