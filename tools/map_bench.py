"""Timing of yh_map_11point: the reference's own scratch shape (P=1024, G=32, C=10)
and a dataset-sized call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
def run(P, G, C, iters):
    g = torch.Generator().manual_seed(0)
    gb = torch.rand(G, 2, generator=g) * 400
    gt = torch.cat([gb, gb + 10 + torch.rand(G, 2, generator=g) * 40, torch.randint(0, C, (G, 1), generator=g).float()], 1).to(dev)
    src = torch.randint(0, G, (P,), generator=g)
    pred = torch.cat([gt.cpu()[src, :4] + torch.randn(P, 4, generator=g) * 6, torch.rand(P, 1, generator=g),
                      torch.randint(0, C, (P, 1), generator=g).float()], 1).to(dev)
    ws = torch.empty(lib.yh_map_workspace_bytes(P, G), dtype=torch.uint8, device=dev)
    ap = torch.empty(C, dtype=torch.float64, device=dev); m = torch.empty((), dtype=torch.float64, device=dev)
    def f(): _lib.check(lib.yh_map_11point(pred.data_ptr(), gt.data_ptr(), P, G, C, 0.5, ap.data_ptr(), m.data_ptr(), ws.data_ptr(), ws.numel(), st))
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    a.record()
    t0 = time.perf_counter()
    for _ in range(iters): f()
    host = 1e6 * (time.perf_counter() - t0) / iters
    b.record(); torch.cuda.synchronize()
    print(f"P={P} G={G} C={C}: {1e3 * a.elapsed_time(b) / iters:.1f} us per call (host enqueue {host:.1f} us per call), mAP {m.item():.4f}")
cases = [(1024, 32, 10, 200), (16384, 512, 20, 100), (100_000, 5_000, 20, 20), (1_000_000, 50_000, 20, 5)]
for i, cse in enumerate(cases):
    if len(sys.argv) < 2 or int(sys.argv[1]) == i:
        run(*cse)
