"""Per-wavefront view of the stress-shape kernel's pair phase (probe build): when each of an
image's eight wavefronts enters and leaves the row queue, how many groups of 64 rows it drew and
how many loop iterations (longest row of each group) it ran."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, MODE_DENSE
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_BATCH", "1024"))
S, BX, C, IMG = 14, 5, 80, 448
head = YOLOv1Head(IMG, S, BX, C)
pred = synth.make_inputs(B, S, BX, C, IMG)[0].to(dev)
stamps = torch.zeros(B * 16 + B * 32, dtype=torch.int64, device=dev)
for _ in range(5): head.postprocess(pred, 0.001, 0.5, MODE_DENSE)
os.environ["YH_NMS_STAMPS"] = str(stamps.data_ptr())
stamps.zero_()
head.postprocess(pred, 0.001, 0.5, MODE_DENSE)
torch.cuda.synchronize()
s = stamps.cpu()
ph = s[:B * 16].view(B, 16).double()
wi = s[B * 16:].view(B, 8, 4)
cu = (wi[:, 0, 2] >> 32)
wi[:, :, 2] &= 0xffffffff
w = wi.double()
dur = (w[:, :, 1] - w[:, :, 0]) / 100.0
print("batch", B)
print("pair phase by stamps (10 -> 11): mean %.2f us" % ((ph[:, 11] - ph[:, 10]).mean().item() / 100.0))
print("per wavefront: in the queue loop mean %.2f us, max over an image's wavefronts mean %.2f us" % (dur.mean().item(), dur.max(1).values.mean().item()))
print("  entry spread within an image: mean %.2f us" % ((w[:, :, 0].max(1).values - w[:, :, 0].min(1).values).mean().item() / 100.0))
print("groups per wavefront: mean %.2f  max %.0f;  per image %.1f" % (w[:, :, 2].mean().item(), w[:, :, 2].max().item(), w[:, :, 2].sum(1).mean().item()))
print("iterations per wavefront: mean %.2f  max %.0f;  per image %.1f" % (w[:, :, 3].mean().item(), w[:, :, 3].max().item(), w[:, :, 3].sum(1).mean().item()))
it = w[:, :, 3].clamp(min=1)
print("us per iteration (loop time / iterations): mean %.3f" % (dur / it).mean().item())
for b in (0, 1, B // 2):
    print("image", b, "wavefronts (groups, iterations, us):", [(int(w[b, k, 2]), int(w[b, k, 3]), round(dur[b, k].item(), 2)) for k in range(8)])
imax = w[:, :, 3].max(1).values
print("iterations: max over an image's wavefronts, mean %.1f; mean over wavefronts %.1f" % (imax.mean().item(), w[:, :, 3].mean().item()))

# which CU each image ran on, and how the arrival of the data spreads inside a CU and across CUs
memdone = (ph[:, 2] - ph[:, 0].min()) / 100.0
end = (ph[:, 15] - ph[:, 0].min()) / 100.0
ids = cu.unique()
print("CUs used:", len(ids), " images per CU: min %d max %d" % (min((cu == i).sum().item() for i in ids), max((cu == i).sum().item() for i in ids)))
within = torch.stack([memdone[cu == i].max() - memdone[cu == i].min() for i in ids])
means = torch.stack([memdone[cu == i].mean() for i in ids])
print("memory phase done: spread inside a CU mean %.2f us (max %.2f); CU means from %.2f to %.2f us" % (within.mean().item(), within.max().item(), means.min().item(), means.max().item()))
for i in ids[:4].tolist() + ids[-2:].tolist():
    sel = (cu == i).nonzero().flatten()
    print("  CU %4x: blocks %s  data at %s  end %s" % (i, sel.tolist(), [round(x, 1) for x in memdone[sel].tolist()], [round(x, 1) for x in end[sel].tolist()]))
