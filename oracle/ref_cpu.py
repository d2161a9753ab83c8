"""CPU oracle for the YOLOv1 detection head -- TEST INFRASTRUCTURE ONLY.

This module is a CPU (PyTorch fp32 / numpy fp32) restatement of the reference's
head algorithm.  It exists to check the HIP kernels.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``yolo_amd`` never does and has no CPU fallback.

Parity status: the reference holds no tests or golden vectors for this path.
Every primitive below that the reference can execute is pinned by fixtures
generated from the reference itself (``oracle/make_golden.py`` ->
``tests/golden/*.npz``).  The reference's ``get_loss`` and its three loss
terms raise as committed; they are pinned by running the reference's own
composition lines with the helper substitutions the repairs R1-R7 below imply
(fixture g6: coordinate, confidence, class terms and the total -- the patches
are listed in make_golden.py).  What no fixture can pin is the choice of the
repairs themselves.  NMS has no counterpart in the reference at all ->
"parity unpinned" for the NMS keep rule.

Repairs relative to the reference source (reference: /root/reference):
  R1  the decode is batched: normalised xywh has shape (B, S*S, Bx, 4) with
      [b, i*S+j, k, :] = sigmoid(out[b, 5k:5k+4, i, j]) -- the shape that
      model.py:226 indexes and the cell order model.py:247-250 uses.
  R2  grid offsets follow denormalize_xywh's view (model.py:171-180): the FIRST
      spatial axis carries x, the second y (same as get_object_mask,
      model.py:219-221).  The box-major ``cat`` of model.py:165 is dropped.
  R3  IoU is taken per cell: predictor k of a cell against that cell's single
      ground-truth box, both decoded by denormalize_xywh + xywh_to_ltrb
      (model.py:260-263); arithmetic identical to eval.py:53-65.
  R4  responsible predictor = argmax_k IoU, first index on ties (torch.max,
      model.py:210), AND-ed with obj_mask (the keyword model.py:308 passes).
  R5  gt_norm_xywh / obj_mask are passed into the confidence and class terms
      as arguments (model.py:261 and :291 read undefined names).
  R6  IoU stays attached to the autograd graph (model.py:263-269 neither
      detaches it nor wraps it in no_grad).
  R7  the debug print of model.py:316 is dropped.
Nothing else changes: lambda_coord = 5, lambda_noobj = 0.5, ``** 0.5``,
softmax-MSE, division by the batch size.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# box geometry -- eval.py:22-65
# ----------------------------------------------------------------------------
def get_area(ltrb: torch.Tensor) -> torch.Tensor:
    """eval.py:22-32: clip(r-l, 0) * clip(b-t, 0); an inverted box has area 0."""
    return torch.clip(ltrb[..., 2] - ltrb[..., 0], min=0) * torch.clip(
        ltrb[..., 3] - ltrb[..., 1], min=0
    )


def get_intersection_area(ltrb1: torch.Tensor, ltrb2: torch.Tensor) -> torch.Tensor:
    """eval.py:35-50: pairwise (N, M) intersection area."""
    l = torch.maximum(ltrb1[..., 0][..., None], ltrb2[..., 0][None, ...])
    t = torch.maximum(ltrb1[..., 1][..., None], ltrb2[..., 1][None, ...])
    r = torch.minimum(ltrb1[..., 2][..., None], ltrb2[..., 2][None, ...])
    b = torch.minimum(ltrb1[..., 3][..., None], ltrb2[..., 3][None, ...])
    return torch.clip(r - l, min=0) * torch.clip(b - t, min=0)


def get_iou(ltrb1: torch.Tensor, ltrb2: torch.Tensor) -> torch.Tensor:
    """eval.py:53-65: pairwise (N, M) IoU = inter / (a1 + a2 - inter), 0/0 -> 0."""
    a1 = get_area(ltrb1)
    a2 = get_area(ltrb2)
    inter = get_intersection_area(ltrb1, ltrb2)
    union = a1[..., None] + a2[None, ...] - inter
    return torch.where(union == 0, 0.0, inter / union)


def get_smallest_enclosing_area(bbox1: torch.Tensor, bbox2: torch.Tensor) -> torch.Tensor:
    """eval.py:68-73."""
    l = torch.minimum(bbox1[:, 0][:, None], bbox2[:, 0][None, :])
    t = torch.minimum(bbox1[:, 1][:, None], bbox2[:, 1][None, :])
    r = torch.maximum(bbox1[:, 2][:, None], bbox2[:, 2][None, :])
    b = torch.maximum(bbox1[:, 3][:, None], bbox2[:, 3][None, :])
    return torch.clip(r - l, min=0) * torch.clip(b - t, min=0)


def get_giou(bbox1: torch.Tensor, bbox2: torch.Tensor) -> torch.Tensor:
    """eval.py:68-83: pairwise GIoU = IoU - (C - union)/C with C the area of the
    smallest enclosing box; -1 where C == 0."""
    a1 = get_area(bbox1)
    a2 = get_area(bbox2)
    inter = get_intersection_area(bbox1, bbox2)
    union = a1[:, None] + a2[None, :] - inter
    l = torch.minimum(bbox1[:, 0][:, None], bbox2[:, 0][None, :])
    t = torch.minimum(bbox1[:, 1][:, None], bbox2[:, 1][None, :])
    r = torch.maximum(bbox1[:, 2][:, None], bbox2[:, 2][None, :])
    b = torch.maximum(bbox1[:, 3][:, None], bbox2[:, 3][None, :])
    c = torch.clip(r - l, min=0) * torch.clip(b - t, min=0)
    iou = torch.where(union == 0, 0, inter / union)
    return torch.where(c == 0, -1, iou - ((c - union) / c))


def iou_elementwise(ltrb1: torch.Tensor, ltrb2: torch.Tensor) -> torch.Tensor:
    """Same arithmetic and operation order as get_iou (eval.py:61-65), applied
    to broadcast-compatible (..., 4) tensors instead of all pairs (repair R3)."""
    a1 = get_area(ltrb1)
    a2 = get_area(ltrb2)
    l = torch.maximum(ltrb1[..., 0], ltrb2[..., 0])
    t = torch.maximum(ltrb1[..., 1], ltrb2[..., 1])
    r = torch.minimum(ltrb1[..., 2], ltrb2[..., 2])
    b = torch.minimum(ltrb1[..., 3], ltrb2[..., 3])
    inter = torch.clip(r - l, min=0) * torch.clip(b - t, min=0)
    union = a1 + a2 - inter
    return torch.where(union == 0, 0.0, inter / union)


# ----------------------------------------------------------------------------
# grid decode -- model.py:150-188
# ----------------------------------------------------------------------------
def get_norm_xywh(out: torch.Tensor, n_bboxes: int) -> torch.Tensor:
    """model.py:157-165 batched per R1: (B, 5*Bx+C, S, S) -> (B, S*S, Bx, 4)."""
    bsz, _, s, _ = out.shape
    boxes = [
        torch.sigmoid(out[:, 5 * k: 5 * k + 4]).permute(0, 2, 3, 1).reshape(bsz, s * s, 4)
        for k in range(n_bboxes)
    ]
    return torch.stack(boxes, dim=2)


def denormalize_xywh(norm_xywh: torch.Tensor, n_cells: int, img_size: int) -> torch.Tensor:
    """model.py:167-183 with R2: x = nx*cell + cell*i, y = ny*cell + cell*j for
    cell index i*S+j; w, h scale by img_size.  Input (B, S*S, K, 4)."""
    cell_size = img_size // n_cells                      # model.py:114
    xywh = norm_xywh.clone()
    xywh[..., 0:2] *= cell_size                          # model.py:170
    grid = torch.arange(0, img_size, cell_size, device=norm_xywh.device)[:n_cells]
    cells = torch.arange(n_cells * n_cells, device=norm_xywh.device)
    x_off = grid[cells // n_cells].to(xywh.dtype)        # first spatial axis -> x
    y_off = grid[cells % n_cells].to(xywh.dtype)         # second spatial axis -> y
    xywh[..., 0] += x_off[None, :, None]                 # model.py:175-177
    xywh[..., 1] += y_off[None, :, None]                 # model.py:178-180
    xywh[..., 2:4] *= img_size                           # model.py:182
    return xywh


def xywh_to_ltrb(xywh: torch.Tensor, img_size: int) -> torch.Tensor:
    """model.py:150-155: one-sided clips only."""
    l = torch.clip(xywh[..., 0] - (xywh[..., 2] / 2), min=0)
    t = torch.clip(xywh[..., 1] - (xywh[..., 3] / 2), min=0)
    r = torch.clip(xywh[..., 0] + (xywh[..., 2] / 2), max=img_size)
    b = torch.clip(xywh[..., 1] + (xywh[..., 3] / 2), max=img_size)
    return torch.stack([l, t, r, b], dim=-1)


def get_ltrb(out: torch.Tensor, n_cells: int, n_bboxes: int, img_size: int) -> torch.Tensor:
    """model.py:185-188 batched: (B, D, S, S) -> (B, S*S, Bx, 4) pixels."""
    return xywh_to_ltrb(
        denormalize_xywh(get_norm_xywh(out, n_bboxes), n_cells, img_size), img_size
    )


def get_confidence(out: torch.Tensor, n_bboxes: int) -> torch.Tensor:
    """model.py:242-250 for general Bx: sigmoid of channels 4, 9, ... -> (B, S*S, Bx, 1)."""
    bsz, _, s, _ = out.shape
    idx = [5 * k + 4 for k in range(n_bboxes)]
    return torch.sigmoid(out[:, idx]).permute(0, 2, 3, 1).reshape(bsz, s * s, n_bboxes)[..., None]


def get_classification_prob(out: torch.Tensor, n_bboxes: int, n_classes: int) -> torch.Tensor:
    """model.py:278-286: softmax over the class channels -> (B, S*S, 1, C)."""
    bsz, _, s, _ = out.shape
    logits = out[:, 5 * n_bboxes: 5 * n_bboxes + n_classes]
    logits = logits.permute(0, 2, 3, 1).reshape(bsz, s * s, n_classes)
    return F.softmax(logits[:, :, None, :], dim=-1)


# ----------------------------------------------------------------------------
# loss -- model.py:190-317
# ----------------------------------------------------------------------------
def loss_terms(
    out: torch.Tensor,
    gt_norm_xywh: torch.Tensor,
    gt_cls_prob: torch.Tensor,
    obj_mask: torch.Tensor,
    n_cells: int = 7,
    n_bboxes: int = 2,
    n_classes: int = 20,
    img_size: int = 448,
    coord_coeff: float = 5.0,
    noobj_coeff: float = 0.5,
):
    """The three loss SUMS (not divided by the batch) and the responsibility mask.

    out (B, 5Bx+C, S, S); gt_norm_xywh (B, S*S, 1, 4); gt_cls_prob (B, S*S, 1, C);
    obj_mask (B, S*S, 1, 1) bool.
    """
    # model.py:208 / :260 -- predicted boxes in pixels
    pred_norm_xywh = get_norm_xywh(out, n_bboxes)                        # (B,N,Bx,4)
    pred_ltrb = xywh_to_ltrb(denormalize_xywh(pred_norm_xywh, n_cells, img_size), img_size)
    # model.py:261-262 -- ground truth through the same decode
    gt_ltrb = xywh_to_ltrb(denormalize_xywh(gt_norm_xywh, n_cells, img_size), img_size)
    # model.py:263 (R3, R6): IoU stays in the graph
    iou = iou_elementwise(pred_ltrb, gt_ltrb)[..., None]                 # (B,N,Bx,1)

    # model.py:210-211 (R4)
    _, max_iou_idx = torch.max(iou[..., 0], dim=2, keepdim=False)
    resp_mask = F.one_hot(max_iou_idx, num_classes=n_bboxes).bool()[..., None]
    resp_mask = resp_mask & obj_mask.bool()                              # (B,N,Bx,1)

    # model.py:224-240
    gt_xy = gt_norm_xywh[:, :, :, 0:2].repeat(1, 1, n_bboxes, 1)
    xy_loss = torch.sum(
        resp_mask * F.mse_loss(pred_norm_xywh[..., 0:2], gt_xy, reduction="none")
    )
    gt_wh = gt_norm_xywh[:, :, :, 2:4].repeat(1, 1, n_bboxes, 1)
    wh_loss = torch.sum(
        resp_mask * F.mse_loss(pred_norm_xywh[..., 2:4] ** 0.5, gt_wh ** 0.5, reduction="none")
    )
    coord_loss = coord_coeff * (xy_loss + wh_loss)

    # model.py:252-276
    pred_conf = get_confidence(out, n_bboxes)
    obj_conf_loss = torch.sum(resp_mask * F.mse_loss(pred_conf, iou, reduction="none"))
    noobj_conf_loss = noobj_coeff * torch.sum(
        (~resp_mask) * F.mse_loss(pred_conf, torch.zeros_like(pred_conf), reduction="none")
    )
    conf_loss = obj_conf_loss + noobj_conf_loss

    # model.py:288-294 (R5)
    pred_cls_prob = get_classification_prob(out, n_bboxes, n_classes)
    cls_loss = torch.sum(
        obj_mask.bool().repeat(1, 1, 1, n_classes)
        * F.mse_loss(pred_cls_prob, gt_cls_prob, reduction="none")
    )
    return coord_loss, conf_loss, cls_loss, resp_mask


def get_loss(out, gt_norm_xywh, gt_cls_prob, obj_mask, global_batch=None, **hp):
    """model.py:296-317: (coord + conf + cls) / B.  ``global_batch`` replaces
    out.size(0) when ``out`` is one shard of a larger batch."""
    coord, conf, cls, _ = loss_terms(out, gt_norm_xywh, gt_cls_prob, obj_mask, **hp)
    denom = out.size(0) if global_batch is None else global_batch
    return (coord + conf + cls) / denom


def loss_and_grad(out, gt_norm_xywh, gt_cls_prob, obj_mask, global_batch=None,
                  dtype=torch.float32, **hp):
    """Loss sums, total/B and d(total/B)/d(out) by autograd (train.py:93)."""
    out = out.detach().to(dtype).clone().requires_grad_(True)
    gx = gt_norm_xywh.to(dtype)
    gc = gt_cls_prob.to(dtype)
    coord, conf, cls, resp = loss_terms(out, gx, gc, obj_mask, **hp)
    denom = out.size(0) if global_batch is None else global_batch
    loss = (coord + conf + cls) / denom
    loss.backward()
    return {
        "coord": coord.detach(), "conf": conf.detach(), "cls": cls.detach(),
        "total": (coord + conf + cls).detach(), "loss": loss.detach(),
        "grad": out.grad.detach(), "resp": resp.detach(),
    }


# ----------------------------------------------------------------------------
# target encoding -- data.py:173-194, model.py:213-222
# ----------------------------------------------------------------------------
def encode_targets(gt_ltrb, gt_label, gt_count, n_cells=7, n_classes=20, img_size=448):
    """Per-image, per-box loop with the reference's own tensor expressions:
    ltrb_to_xywh (data.py:173-183), normalize_xywh (data.py:185-194), the cell
    index of get_object_mask (model.py:217-221: [x // cell_size, y // cell_size],
    first axis = x).  One box per cell (data.py:32): a later box overwrites an
    earlier one.  Indices are clamped into the grid (the reference would raise).
    gt_ltrb (B, M, 4), gt_label (B, M), gt_count (B,)."""
    bsz = gt_ltrb.shape[0]
    n = n_cells * n_cells
    cell_size = img_size // n_cells
    gt_norm_xywh = torch.zeros(bsz, n, 1, 4)
    gt_cls_prob = torch.zeros(bsz, n, 1, n_classes)
    obj_mask = torch.zeros(bsz, n, 1, 1, dtype=torch.bool)
    for b in range(bsz):
        for m in range(int(gt_count[b])):
            ltrb = gt_ltrb[b, m].to(torch.float32)
            xywh = torch.stack([(ltrb[0] + ltrb[2]) / 2, (ltrb[1] + ltrb[3]) / 2,
                                ltrb[2] - ltrb[0], ltrb[3] - ltrb[1]], dim=-1)
            norm = torch.stack([xywh[0] % cell_size / cell_size, xywh[1] % cell_size / cell_size,
                                xywh[2] / img_size, xywh[3] / img_size], dim=-1)
            xc = min(max(int((xywh[0] // cell_size).long()), 0), n_cells - 1)
            yc = min(max(int((xywh[1] // cell_size).long()), 0), n_cells - 1)
            cell = xc * n_cells + yc
            gt_norm_xywh[b, cell, 0] = norm
            gt_cls_prob[b, cell, 0] = 0.0
            lab = int(gt_label[b, m])
            if 0 <= lab < n_classes:
                gt_cls_prob[b, cell, 0, lab] = 1.0
            obj_mask[b, cell, 0, 0] = True
    return gt_norm_xywh, gt_cls_prob, obj_mask


# ----------------------------------------------------------------------------
# post-processing -- model.py:379-394
# ----------------------------------------------------------------------------
def get_confidence_mask(out, conf_thresh, n_bboxes=2):
    """model.py:379-382: max over the predictors of conf >= thresh (inclusive)."""
    max_conf, _ = torch.max(get_confidence(out, n_bboxes), dim=2, keepdim=True)
    return max_conf >= conf_thresh


def get_max_confidence_ltrb(out, n_cells=7, n_bboxes=2, img_size=448):
    """model.py:384-390 on the batched decode: ltrb of the higher-confidence
    predictor per cell (torch.max -> first index on ties)."""
    pred_ltrb = get_ltrb(out, n_cells, n_bboxes, img_size)
    _, max_conf_idx = torch.max(get_confidence(out, n_bboxes), dim=2, keepdim=True)
    return torch.gather(pred_ltrb, dim=2, index=max_conf_idx.repeat(1, 1, 1, 4))


def get_classification_index(out, n_bboxes=2, n_classes=20):
    """model.py:392-394: argmax of the softmax."""
    return torch.argmax(get_classification_prob(out, n_bboxes, n_classes), dim=-1, keepdim=True)


def decode_candidates(out, mode, n_cells=7, n_bboxes=2, n_classes=20, img_size=448):
    """Candidate boxes for NMS, all as (B, S*S*Bx, ...) with flat index
    cell*Bx + k.

    mode 0 ("reference", model.py:379-394): per cell only the higher-confidence
    predictor is a candidate; the other slots of the cell are marked absent.
    mode 1 ("dense"): every predictor is a candidate.
    score = confidence, class = argmax softmax of the cell (model.py:392-394).
    Returns ltrb f32 (B,N,4), score f32 (B,N), cls int64 (B,N), present bool (B,N).
    """
    bsz = out.size(0)
    n = n_cells * n_cells
    ltrb = get_ltrb(out, n_cells, n_bboxes, img_size)                     # (B,n,Bx,4)
    conf = get_confidence(out, n_bboxes)[..., 0]                          # (B,n,Bx)
    cls = get_classification_index(out, n_bboxes, n_classes)[:, :, 0, 0]  # (B,n)
    cls = cls[:, :, None].repeat(1, 1, n_bboxes)
    if mode == 0:
        _, best = torch.max(conf, dim=2, keepdim=True)
        present = torch.zeros_like(conf, dtype=torch.bool).scatter_(2, best, True)
    else:
        present = torch.ones_like(conf, dtype=torch.bool)
    return (
        ltrb.reshape(bsz, n * n_bboxes, 4).contiguous(),
        conf.reshape(bsz, n * n_bboxes).contiguous(),
        cls.reshape(bsz, n * n_bboxes).contiguous(),
        present.reshape(bsz, n * n_bboxes).contiguous(),
    )


def _iou_one_to_many(box: np.ndarray, others: np.ndarray) -> np.ndarray:
    """eval.py:53-65 in numpy fp32 (every operation rounds to fp32, no fused
    multiply-add): IoU of one box against (M, 4) boxes."""
    f = np.float32
    zero = f(0)
    a1 = np.maximum(box[2] - box[0], zero) * np.maximum(box[3] - box[1], zero)
    a2 = np.maximum(others[:, 2] - others[:, 0], zero) * np.maximum(others[:, 3] - others[:, 1], zero)
    l = np.maximum(box[0], others[:, 0])
    t = np.maximum(box[1], others[:, 1])
    r = np.minimum(box[2], others[:, 2])
    b = np.minimum(box[3], others[:, 3])
    inter = np.maximum(r - l, zero) * np.maximum(b - t, zero)
    union = (a1 + a2) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union
    return np.where(union == 0, zero, iou).astype(f)


def nms_image(ltrb, score, cls, present, conf_thresh, iou_thresh):
    """Confidence threshold + per-class greedy NMS for ONE image (parity
    unpinned: the reference has no NMS; this is the build's specification).

    A candidate is valid when present and score >= conf_thresh (inclusive,
    model.py:382).  Valid candidates are ordered by class ascending, then score
    descending, then flat index ascending.  Walking each class in that order, a
    candidate is kept unless an already-kept candidate of the same class has
    IoU > iou_thresh with it (strict; eval.py:53-65 arithmetic in fp32).
    Returns the kept flat indices in that (class, score, index) order.
    """
    ltrb = np.asarray(ltrb, dtype=np.float32)
    score = np.asarray(score, dtype=np.float32)
    cls = np.asarray(cls).astype(np.int64)
    valid = np.asarray(present, dtype=bool) & (score >= np.float32(conf_thresh))
    idx = np.nonzero(valid)[0]
    if idx.size == 0:
        return np.zeros((0,), dtype=np.int64)
    # lexsort: last key is the primary one
    order = idx[np.lexsort((idx, -score[idx].astype(np.float64), cls[idx]))]
    thr = np.float32(iou_thresh)
    kept = []
    start = 0
    while start < order.size:
        c = cls[order[start]]
        end = start
        while end < order.size and cls[order[end]] == c:
            end += 1
        seg = order[start:end]
        alive = np.ones(seg.size, dtype=bool)
        for p in range(seg.size):
            if not alive[p]:
                continue
            kept.append(seg[p])
            if p + 1 < seg.size:
                iou = _iou_one_to_many(ltrb[seg[p]], ltrb[seg[p + 1:]])
                alive[p + 1:] &= ~(iou > thr)
        start = end
    return np.asarray(kept, dtype=np.int64)


def nms_batch(ltrb, score, cls, present, conf_thresh, iou_thresh):
    """nms_image over a batch; returns (dets (B, N, 6) zero-padded rows
    l,t,r,b,score,class as eval.py:89-91 lays them out, keep_count (B,),
    keep_idx (B, N) padded with -1)."""
    ltrb = np.asarray(ltrb, dtype=np.float32)
    score = np.asarray(score, dtype=np.float32)
    cls = np.asarray(cls)
    present = np.asarray(present)
    bsz, n = score.shape
    dets = np.zeros((bsz, n, 6), dtype=np.float32)
    count = np.zeros((bsz,), dtype=np.int32)
    keep_idx = np.full((bsz, n), -1, dtype=np.int32)
    for b in range(bsz):
        kept = nms_image(ltrb[b], score[b], cls[b], present[b], conf_thresh, iou_thresh)
        k = kept.size
        count[b] = k
        keep_idx[b, :k] = kept
        dets[b, :k, 0:4] = ltrb[b, kept]
        dets[b, :k, 4] = score[b, kept]
        dets[b, :k, 5] = cls[b, kept].astype(np.float32)
    return dets, count, keep_idx


def postprocess(out, conf_thresh, iou_thresh, mode=0, **hp):
    """decode_candidates + nms_batch: the CPU statement of the whole
    inference-side path."""
    ltrb, score, cls, present = decode_candidates(out, mode, **hp)
    return nms_batch(ltrb.numpy(), score.numpy(), cls.numpy(), present.numpy(),
                     conf_thresh, iou_thresh)


# ----------------------------------------------------------------------------
# 11-point mAP -- eval.py:86-149, quirks included
# ----------------------------------------------------------------------------
def true_positive_flags(pred, gt, iou_thresh):
    """What eval.py:96-119 computes, for all classes at once.  A prediction is a
    true positive of ITS class when any ground truth of that class has
    8 * IoU >= iou_thresh (the stray ``iou *= 8`` of eval.py:99 is kept; the
    matched flags of eval.py:109-110 are never set, so matching is not one-to-one)."""
    pred = torch.as_tensor(pred, dtype=torch.float32)
    gt = torch.as_tensor(gt, dtype=torch.float32)
    iou = get_iou(gt[:, :4], pred[:, :4])
    iou *= 8
    same = gt[:, 4][:, None] == pred[:, 5][None, :]
    return (same & (iou >= iou_thresh)).any(dim=0).numpy()


def get_ap(pred, gt, cls_idx, iou_thresh, tp=None):
    """eval.py:121-142 for one class: precision counts every prediction of every
    class (eval.py:124), recall divides by ALL ground truths (eval.py:125)."""
    pred = torch.as_tensor(pred, dtype=torch.float32)
    gt = torch.as_tensor(gt, dtype=torch.float32)
    if tp is None:
        tp = true_positive_flags(pred, gt, iou_thresh)
    is_true = tp & (pred[:, 5].numpy() == cls_idx)
    # torch.argsort(descending=True) (eval.py:121) counts a NaN as larger than every number:
    # NaN confidences come first.  Ties (and NaNs among themselves): the lower index first
    # (torch.argsort leaves them unspecified)
    conf = pred[:, 4].numpy().astype(np.float64)
    order = np.lexsort((np.arange(pred.shape[0]), np.where(np.isnan(conf), -np.inf, -conf)))
    accum_tp = np.cumsum(is_true[order])
    with np.errstate(divide="ignore", invalid="ignore"):
        accum_prec = accum_tp / np.arange(1, pred.shape[0] + 1)
        accum_recall = accum_tp / gt.shape[0]
    precs = []
    for recall in np.linspace(0, 1, 11):
        sel = accum_prec[accum_recall >= recall]
        precs.append(0 if not np.any(sel) else sel.max())
    return np.sum(precs) / 11


def get_map(pred, gt, n_classes, iou_thresh=0.5):
    """eval.py:145-149."""
    tp = true_positive_flags(pred, gt, iou_thresh)
    sum_ap = 0
    for cls_idx in range(n_classes):
        sum_ap += get_ap(pred, gt, cls_idx, iou_thresh, tp)
    return sum_ap / n_classes
