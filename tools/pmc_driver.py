"""Driver for the rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs).

Launch order: two calibration copies with known byte counts -- a 16-byte-per-lane
copy and a 4-byte-per-lane copy (source offset by one float so PyTorch's copy
kernel cannot vectorise) -- then the loss kernel (the pipelined step the benchmark
runs), the decode+NMS kernel (S*S rows per image, as the benchmark calls it) and the
one-kernel step (head_step_kernel) of the benchmark workload (batch 4096), each over a
rotation of buffer sets larger than twice the 256 MiB memory-side cache, so every
launch meets its data in HBM, as in bench.py's `beyond_memory_side_cache` leg."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import synth, MODE_REFERENCE, _lib
dev = torch.device("cuda:0")
B = 4096
NSETS = int(os.environ.get("PMC_SETS", "17"))
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
base = [t.to(dev) for t in synth.make_inputs(B)]
sets = []
for _ in range(NSETS):
    pred, gx, gc, om = (t.clone() for t in base)
    sets.append((pred, gx, gc, om.view(torch.uint8), torch.empty_like(pred),
                 torch.empty(B, 49, 6, device=dev), torch.empty(B, dtype=torch.int32, device=dev)))
n = 16 * 1024 * 1024                     # 64 MiB of floats
x = torch.randn(n + 4, device=dev); y = torch.empty(n + 4, device=dev)
torch.cuda.synchronize()
y[:n].copy_(x[:n]); torch.cuda.synchronize()          # calibration 1: 16 B per lane
y[:n].copy_(x[1:n + 1]); torch.cuda.synchronize()     # calibration 2: 4 B per lane
wsp = torch.zeros(lib.yh_workspace_bytes(_lib.YH_OP_LOSS_PIPELINED, B, 7, 2, 20), dtype=torch.uint8, device=dev)
ring = torch.zeros(64, 4, device=dev); cur = torch.zeros(1, dtype=torch.int32, device=dev)
for i in range(2 * NSETS):
    p, gx, gc, om8, grad, dets, cnt = sets[i % NSETS]
    _lib.check(lib.yh_loss_fwd_bwd_pipelined(p.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(),
                                             None, ring.data_ptr(), 64, cur.data_ptr(), wsp.data_ptr(), wsp.numel(), i & 1,
                                             B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st))
torch.cuda.synchronize()
for i in range(2 * NSETS):
    p, gx, gc, om8, grad, dets, cnt = sets[i % NSETS]
    _lib.check(lib.yh_decode_nms_rows(p.data_ptr(), dets.data_ptr(), cnt.data_ptr(), None, MODE_REFERENCE, 0.5, 0.5,
                                      49, B, 7, 2, 20, 448, st))
torch.cuda.synchronize()
for i in range(2 * NSETS):
    p, gx, gc, om8, grad, dets, cnt = sets[i % NSETS]
    _lib.check(lib.yh_head_step_pipelined(p.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(),
                                          dets.data_ptr(), cnt.data_ptr(), None, None, ring.data_ptr(), 64, cur.data_ptr(),
                                          wsp.data_ptr(), wsp.numel(), i & 1, MODE_REFERENCE, 0.5, 0.5, 49,
                                          B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st))
torch.cuda.synchronize()
print("calibration bytes per copy:", 4 * n, "obj cells:", int(base[3].sum()), "buffer sets:", NSETS)
