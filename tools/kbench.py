"""Back-to-back HIP-event timing of the three kernels of the step (batch 4096)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_amd import YOLOv1Head, synth, MODE_REFERENCE, MODE_DENSE, _lib
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_BATCH", "4096"))
head = YOLOv1Head()
pred, gx, gc, om = (t.to(dev) for t in synth.make_inputs(B))
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
ws = head._get_workspace(dev, lib.yh_workspace_bytes(0, B, 7, 2, 20))
grad = torch.empty_like(pred); om8 = om.view(torch.uint8)
dets = torch.empty(B, 98, 6, device=dev); cnt = torch.empty(B, dtype=torch.int32, device=dev); terms = torch.empty(5, device=dev)
mode = int(os.environ.get("KB_MODE", "0"))
conf = float(os.environ.get("KB_CONF", "0.5"))
def k_loss(): lib.yh_loss_partials(pred.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st)
from yolo_amd import LossLog
plog = LossLog(dev, 64)
wsp = torch.zeros(lib.yh_workspace_bytes(5, B, 7, 2, 20), dtype=torch.uint8, device=dev)
ph = [0]
def k_pipe():
    ph[0] ^= 1
    lib.yh_loss_fwd_bwd_pipelined(pred.data_ptr(), gx.data_ptr(), gc.data_ptr(), om8.data_ptr(), grad.data_ptr(), None, plog.ring.data_ptr(), 64, plog.cursor.data_ptr(), wsp.data_ptr(), wsp.numel(), ph[0], B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st)
def k_two(): k_loss(); k_fin()
def k_fin(): lib.yh_loss_finalize(ws.data_ptr(), ws.numel(), B, 7, 1.0 / B, terms.data_ptr(), None, 0, None, st)
def k_nms(): lib.yh_decode_nms(pred.data_ptr(), dets.data_ptr(), cnt.data_ptr(), None, mode, conf, 0.5, B, 7, 2, 20, 448, st)
def b2b(fn, iters=300):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
lt, lb, ct = (t.to(dev) for t in synth.make_gt_boxes(B, seed=1))
lb = lb.to(torch.int32); ct = ct.to(torch.int32)
ex = torch.empty(B, 49, 1, 4, device=dev); ec = torch.empty(B, 49, 1, 20, device=dev); em = torch.empty(B, 49, 1, 1, dtype=torch.uint8, device=dev)
def k_enc(): lib.yh_encode_targets(lt.data_ptr(), lb.data_ptr(), ct.data_ptr(), ex.data_ptr(), ec.data_ptr(), em.data_ptr(), B, lt.shape[1], 7, 20, 448, st)
def k_encloss():
    k_enc()
    lib.yh_loss_partials(pred.data_ptr(), ex.data_ptr(), ec.data_ptr(), em.data_ptr(), grad.data_ptr(), ws.data_ptr(), ws.numel(), B, 7, 2, 20, 448, 5.0, 0.5, 1.0 / B, st)
which = sys.argv[1:] or ["loss", "fin", "nms"]
out = []
for w in which:
    out.append("%s %.2f us" % (w, b2b({"loss": k_loss, "pipe": k_pipe, "two": k_two, "fin": k_fin, "nms": k_nms, "enc": k_enc, "encloss": k_encloss}[w])))
print(os.environ.get("KB_TAG", ""), " | ".join(out))
