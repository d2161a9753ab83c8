"""Random shapes, thresholds, layouts and class crowding through the fused decode+NMS
kernels, each result compared bit for bit with the CPU oracle's NMS of the candidates the
device decoded (the comparison tests/test_nms_gpu.py::test_fused_decode_nms makes on its
fixed list).  A development aid for changes to nms_kernels.hpp; the oracle is the checker.

    python tools/nms_fuzz.py [configs] [seed]
    NF_SLOTS=65,128 python tools/nms_fuzz.py 500     (slot counts of one kernel variant only)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import ref_cpu
from yolo_amd import MODE_DENSE, MODE_REFERENCE, YOLOv1Head, synth

dev = torch.device("cuda:0")
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
# NF_SLOTS="65,128" keeps the slot count S*S*Bx of every config inside that range
SLOT_LO, SLOT_HI = (int(v) for v in os.environ.get("NF_SLOTS", "1,2048").split(","))
t_start = time.time()
longest = most = 0      # longest class segment and most candidates of an image so far
for it_no in range(n_cfg):
    while True:
        S = int(rng.integers(1, 21)); Bx = int(rng.integers(1, 6)); C = int(rng.choice([1, 2, 3, 7, 20, 33, 80, 100]))
        if SLOT_LO <= S * S * Bx <= SLOT_HI:
            break
    batch = int(rng.integers(1, 4))
    mode = MODE_DENSE if rng.random() < 0.7 else MODE_REFERENCE
    ct = float(rng.choice([0.0, 0.001, 0.1, 0.3, 0.5]))
    iou_t = float(rng.choice([0.05, 0.3, 0.5, 0.7]))
    crowd = float(rng.choice([0.0, 0.0, 0.15, 0.5, 1.0]))
    head = YOLOv1Head(img_size=32 * S, n_cells=S, n_bboxes=Bx, n_classes=C)
    pred = synth.make_pred(batch, S, Bx, C, seed=int(rng.integers(1 << 30)))
    if crowd > 0:
        mask = torch.from_numpy(rng.random((batch, S, S)) < crowd)
        pred[:, 5 * Bx + C // 2][mask] += 30.0
    if rng.random() < 0.3:      # crowded boxes: many IoUs above the threshold
        pred[:, [5 * k + q for k in range(Bx) for q in (2, 3)]] += 1.0
    pred = pred.to(dev)
    ltrb, score, cls, present = head.decode_candidates(pred, mode)
    want = ref_cpu.nms_batch(ltrb.cpu().numpy(), score.cpu().numpy(), cls.cpu().numpy(),
                             present.cpu().numpy(), ct, iou_t)
    valid = (present & (score >= ct)).cpu()
    for b_ in range(batch):
        cl = cls[b_].cpu()[valid[b_]]
        if cl.numel():
            longest = max(longest, int(torch.bincount(cl.long()).max()))
            most = max(most, int(cl.numel()))
    for layout in ("channel-major", "cell-major"):
        x = pred if layout == "channel-major" else pred.contiguous(memory_format=torch.channels_last)
        got = [g.cpu().numpy() for g in head.postprocess(x, ct, iou_t, mode, return_idx=True)]
        ok = (np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
              and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)))
        if not ok:
            print(f"MISMATCH config {it_no}: S={S} Bx={Bx} C={C} batch={batch} mode={mode} conf={ct} iou={iou_t} "
                  f"crowd={crowd} {layout}", flush=True)
            sys.exit(1)
    if it_no % 20 == 19:
        print(f"{it_no + 1} configs equal to the oracle ({time.time() - t_start:.0f} s; up to {most} candidates, "
              f"class segments up to {longest})", flush=True)
print(f"all {n_cfg} configs equal to the oracle")
