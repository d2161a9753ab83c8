"""Where a launch of the one-wavefront-per-image decode+NMS kernel spends its time (probe
build; stamps by the first wavefront of every workgroup), resident and over rotating buffers,
launches back to back."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import synth, _lib
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_BATCH", "4096"))
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
base = synth.make_inputs(B)[0]
nblk = (B + 3) // 4
stamps = torch.zeros(nblk, 16, dtype=torch.int64, device=dev)
cols = [(0, "entry"), (2, "logits in, candidates decoded"), (3, "class table, pairs"), (4, "kept set"), (15, "rows stored (issued)")]
K = int(os.environ.get("KB_B2B", "6"))
MODE = int(os.environ.get("KB_MODE", "0")); CONF = float(os.environ.get("KB_CONF", "0.5"))   # KB_MODE=1: a candidate per predictor
for label, nsets in (("resident", 1), ("rotating", int(os.environ.get("KB_SETS", "18")))):
    sets = [(base.to(dev).clone(), torch.empty(B, 98, 6, device=dev), torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(nsets)]
    def run(i):
        p, d, c = sets[i % nsets]
        _lib.check(lib.yh_decode_nms(p.data_ptr(), d.data_ptr(), c.data_ptr(), None, MODE, CONF, 0.5, B, 7, 2, 20, 448, st))
    for i in range(3 * nsets): run(i)
    torch.cuda.synchronize()
    os.environ["YH_NMS_STAMPS"] = str(stamps.data_ptr())
    acc = torch.zeros(len(cols), dtype=torch.float64); last = 0.0
    N = 2 * nsets
    for i in range(N):
        for q in range(K): run(i * K + q)
        torch.cuda.synchronize()
        s = stamps.cpu().double()[:, [c for c, _ in cols]]
        s = (s - s[:, 0].min()) / 100.0
        acc += s.mean(0); last += s[:, -1].max().item()
    del os.environ["YH_NMS_STAMPS"]
    acc /= N
    print(label, "(%d sets)" % nsets)
    prev = 0.0
    for i, (_, name) in enumerate(cols):
        print("  %-32s at %.2f us (+%.2f)" % (name, acc[i].item(), acc[i].item() - prev)); prev = acc[i].item()
    print("  last workgroup has issued its stores at %.2f us" % (last / N))
    del sets
