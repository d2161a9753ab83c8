"""Host-side target encoding for the YOLOv1 head: the data loader's collate step,
where the boxes still live in host memory (data.py:196-209).  It is not a
fallback for anything on the GPU path -- on the device the same encoding is
``YOLOv1Head.encode_targets`` (yh_encode_targets), and no GPU entry point calls
into this module.

Turns per-image ground-truth boxes (pixel ltrb) and labels into the three
tensors the head's loss consumes:

  gt_norm_xywh (B, S*S, 1, 4) f32   cell-relative centre, image-relative size
  gt_cls_prob  (B, S*S, 1, C) f32   one-hot class
  obj_mask     (B, S*S, 1, 1) bool  a ground-truth centre falls in the cell

The arithmetic follows the reference's ``ltrb_to_xywh`` (data.py:173-183),
``normalize_xywh`` (data.py:185-194) and ``get_object_mask`` (model.py:213-222):
the cell of a box is ``[cx // cell_size, cy // cell_size]`` with the FIRST grid
axis carrying x, flattened as ``x_cell * S + y_cell``.  When several boxes share
a cell the last one in the list wins (one box per cell, data.py:32).
"""
from __future__ import annotations

from typing import Sequence

import torch


def ltrb_to_xywh(ltrb: torch.Tensor) -> torch.Tensor:
    """data.py:173-183: (l, t, r, b) -> (cx, cy, w, h)."""
    return torch.stack(
        [
            (ltrb[..., 0] + ltrb[..., 2]) / 2,
            (ltrb[..., 1] + ltrb[..., 3]) / 2,
            ltrb[..., 2] - ltrb[..., 0],
            ltrb[..., 3] - ltrb[..., 1],
        ],
        dim=-1,
    )


def normalize_xywh(xywh: torch.Tensor, cell_size: int, img_size: int) -> torch.Tensor:
    """data.py:185-194: centre as a fraction of its cell, size as a fraction of the image."""
    return torch.stack(
        [
            xywh[..., 0] % cell_size / cell_size,
            xywh[..., 1] % cell_size / cell_size,
            xywh[..., 2] / img_size,
            xywh[..., 3] / img_size,
        ],
        dim=-1,
    )


def cell_index(xywh: torch.Tensor, n_cells: int, cell_size: int) -> torch.Tensor:
    """Flat cell index x_cell * S + y_cell of each box centre (model.py:217-221)."""
    xc = torch.clamp((xywh[..., 0] // cell_size).long(), 0, n_cells - 1)
    yc = torch.clamp((xywh[..., 1] // cell_size).long(), 0, n_cells - 1)
    return xc * n_cells + yc


def encode_padded(
    ltrb: torch.Tensor,
    label: torch.Tensor,
    count: torch.Tensor,
    n_cells: int = 7,
    n_classes: int = 20,
    img_size: int = 448,
):
    """Vectorised encoder for a padded batch.

    ltrb (B, M, 4) f32 pixels, label (B, M) int64, count (B,) int64 = number of
    real boxes per image (slots >= count are ignored).  Boxes are written in slot
    order, so a later box overwrites an earlier one in the same cell.
    """
    ltrb = ltrb.to(torch.float32)
    bsz, m, _ = ltrb.shape
    n = n_cells * n_cells
    cell_size = img_size // n_cells
    xywh = ltrb_to_xywh(ltrb)
    norm = normalize_xywh(xywh, cell_size, img_size)
    cell = cell_index(xywh, n_cells, cell_size)

    gt_norm_xywh = torch.zeros(bsz, n, 4, dtype=torch.float32)
    gt_cls_prob = torch.zeros(bsz, n, n_classes, dtype=torch.float32)
    obj_mask = torch.zeros(bsz, n, dtype=torch.bool)
    rows = torch.arange(bsz)
    for slot in range(m):
        live = slot < count
        if not bool(live.any()):
            continue
        r = rows[live]
        c = cell[live, slot]
        gt_norm_xywh[r, c] = norm[live, slot]
        gt_cls_prob[r, c] = 0.0
        gt_cls_prob[r, c, label[live, slot]] = 1.0
        obj_mask[r, c] = True
    return (
        gt_norm_xywh.view(bsz, n, 1, 4),
        gt_cls_prob.view(bsz, n, 1, n_classes),
        obj_mask.view(bsz, n, 1, 1),
    )


def encode_targets(
    ltrbs: Sequence[torch.Tensor],
    labels: Sequence[torch.Tensor],
    n_cells: int = 7,
    n_classes: int = 20,
    img_size: int = 448,
):
    """Encoder for the ragged lists ``collate_fn`` yields (data.py:205-209:
    ``annots = {"ltrbs": [...], "labels": [...]}``)."""
    bsz = len(ltrbs)
    if bsz != len(labels):
        raise ValueError("ltrbs and labels must have one entry per image")
    m = max([int(x.shape[0]) for x in ltrbs] + [1])
    ltrb = torch.zeros(bsz, m, 4, dtype=torch.float32)
    label = torch.zeros(bsz, m, dtype=torch.long)
    count = torch.zeros(bsz, dtype=torch.long)
    for b, (box, lab) in enumerate(zip(ltrbs, labels)):
        k = int(box.shape[0])
        if k != int(lab.shape[0]):
            raise ValueError(f"image {b}: {k} boxes but {int(lab.shape[0])} labels")
        count[b] = k
        if k:
            ltrb[b, :k] = box.to(torch.float32).reshape(k, 4)
            label[b, :k] = lab.reshape(k).long()
    if int(label.max()) >= n_classes or int(label.min()) < 0:
        raise ValueError("label out of range")
    return encode_padded(ltrb, label, count, n_cells, n_classes, img_size)
