"""Where a launch of the one-kernel step (head_step_kernel) spends its time: stamps that
thread 0 of every workgroup takes at the phase boundaries (probe build: make -C
yolo_amd/csrc probe; YOLO_AMD_LIB=build/libyolohead_probe.so), over rotating buffer sets
and on one resident set; and the kernel's back-to-back duration in both regimes."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import synth, _lib
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_BATCH", "4096"))
ROWS = int(os.environ.get("KB_ROWS", "49"))
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
raw = ctypes.CDLL(_lib.LIB_PATH)
probe = hasattr(raw, "yh_probe_set_loss_stamps")
if probe:
    raw.yh_probe_set_loss_stamps.argtypes = [ctypes.c_void_p]
ws = torch.empty(lib.yh_workspace_bytes(0, B, 7, 2, 20), dtype=torch.uint8, device=dev)
base = synth.make_inputs(B)
terms = torch.empty(5, device=dev)
def make():
    pred, gx, gc, om = (t.to(dev).clone() for t in base)
    return (pred, gx, gc, om.view(torch.uint8), torch.empty_like(pred), torch.empty(B, ROWS, 6, device=dev),
            torch.empty(B, dtype=torch.int32, device=dev))
nblk = (B + 3) // 4
stamps = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
names = ["entry", "logits in, pass 1 done", "NMS done, rows issued", "barrier", "pass 2 (object cells)",
         "block sums", "tile stored (issued)"]
for label, nsets in (("rotating", int(os.environ.get("KB_SETS", "17"))), ("resident", 1)):
    sets = [make() for _ in range(nsets)]
    def run(i):
        p, gx, gc, om, g, d, c = sets[i % nsets]
        # the two-launch form: the fused kernel, then the (tiny) finalize
        _lib.check(lib.yh_head_step(p.data_ptr(), gx.data_ptr(), gc.data_ptr(), om.data_ptr(), g.data_ptr(),
                                    terms.data_ptr(), d.data_ptr(), c.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                    0, 0.5, 0.5, ROWS, B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st))
    for i in range(3 * nsets): run(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(20 * nsets if nsets > 1 else 300): run(i)
    b.record(); torch.cuda.synchronize()
    n = 20 * nsets if nsets > 1 else 300
    print("%s (%d sets): %.2f us per call (fused kernel + finalize), back to back" % (label, nsets, 1e3 * a.elapsed_time(b) / n))
    if not probe:
        continue
    raw.yh_probe_set_loss_stamps(stamps.data_ptr())
    acc = torch.zeros(7, dtype=torch.float64); spread = []
    N = 3 * nsets
    K = int(os.environ.get("KB_B2B", "3"))      # launches back to back before the one that is read
    for i in range(N):
        for q in range(K): run(i * K + q)
        torch.cuda.synchronize()
        s = stamps.cpu().double()[:, :7]
        s = (s - s[:, 0].min()) / 100.0
        acc += s.mean(0)
        spread.append((s[:, 0].max().item(), s[:, 1].max().item(), s[:, 2].max().item(), s[:, 6].max().item()))
    raw.yh_probe_set_loss_stamps(None)
    acc /= N
    prev = 0.0
    for i in range(7):
        print("  %-28s at %.2f us (+%.2f)" % (names[i], acc[i].item(), acc[i].item() - prev)); prev = acc[i].item()
    for name, col in (("entry", 0), ("logits in", 1), ("rows issued", 2), ("tile issued", 6)):
        q = torch.quantile(s[:, col], torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))
        print("  %-12s min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f us (last launch)" % ((name,) + tuple(q.tolist())))
    sp = torch.tensor(spread).mean(0)
    print("  last workgroup: enters at %.2f, has its logits at %.2f, its rows issued at %.2f, its tile issued at %.2f us" % tuple(sp.tolist()))
    del sets
