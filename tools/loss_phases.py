"""Where a launch of the fused loss kernel spends its time: stamps that thread 0 of every
workgroup takes at the phase boundaries (probe build: make -C yolo_amd/csrc probe;
YOLO_AMD_LIB=build/libyolohead_probe.so), on one resident buffer set and over rotating ones."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import synth, _lib
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_BATCH", "4096"))
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.yh_probe_set_loss_stamps.argtypes = [ctypes.c_void_p]
ws = torch.empty(lib.yh_workspace_bytes(0, B, 7, 2, 20), dtype=torch.uint8, device=dev)
base = synth.make_inputs(B)
def make():
    pred, gx, gc, om = (t.to(dev).clone() for t in base)
    return pred, gx, gc, om.view(torch.uint8), torch.empty_like(pred)
nblk = (B + 3) // 4
stamps = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
names = ["entry", "pass 1 (mask, logits in; no-object cells done)", "barrier", "pass 2 (object cells)",
         "block sums", "tile stored (issued)"]
for label, nsets in (("resident", 1), ("rotating", int(os.environ.get("KB_SETS", "12")))):
    sets = [make() for _ in range(nsets)]
    def run(i):
        p, gx, gc, om, g = sets[i % nsets]
        _lib.check(lib.yh_loss_partials(p.data_ptr(), gx.data_ptr(), gc.data_ptr(), om.data_ptr(), g.data_ptr(),
                                        ws.data_ptr(), ws.numel(), B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st))
    for i in range(3 * nsets): run(i)
    torch.cuda.synchronize()
    raw.yh_probe_set_loss_stamps(stamps.data_ptr())
    acc = torch.zeros(6, dtype=torch.float64); spread = []
    N = 3 * nsets
    K = int(os.environ.get("KB_B2B", "1"))      # launches back to back before the one that is read
    for i in range(N):
        for q in range(K): run(i * K + q)
        torch.cuda.synchronize()
        s = stamps.cpu().double()[:, :6]
        s = (s - s[:, 0].min()) / 100.0
        acc += s.mean(0)
        spread.append((s[:, 0].max().item(), s[:, 1].max().item(), s[:, 5].max().item()))
    raw.yh_probe_set_loss_stamps(None)
    acc /= N
    print(label, "(%d sets)" % nsets)
    prev = 0.0
    for i in range(6):
        print("  %-48s at %.2f us (+%.2f)" % (names[i], acc[i].item(), acc[i].item() - prev)); prev = acc[i].item()
    for name, col in (("entry", 0), ("pass-1 data", 1), ("stores issued", 5)):
        q = torch.quantile(s[:, col], torch.tensor([0.0, 0.01, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))
        print("  %-14s min %.2f  p1 %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f us (last launch)" % ((name,) + tuple(q.tolist())))
    e = s[:, 0]
    print("  entry by workgroup id, mean of each 1/16 of the grid:", " ".join("%.2f" % e[i * len(e) // 16:(i + 1) * len(e) // 16].mean().item() for i in range(16)))
    print("  entry by workgroup id modulo 8 (the XCD a workgroup runs on):", " ".join("%.2f" % e[i::8].mean().item() for i in range(8)))
    print("  stores issued, by id modulo 8:", " ".join("%.2f" % s[i::8, 5].mean().item() for i in range(8)))
    print("  entry deciles:", " ".join("%.2f" % v for v in torch.quantile(e, torch.linspace(0, 1, 11, dtype=torch.float64)).tolist()))
    sp = torch.tensor(spread).mean(0)
    print("  last workgroup: enters at %.2f, has its pass-1 data at %.2f, has issued its stores at %.2f us" % tuple(sp.tolist()))
    del sets
