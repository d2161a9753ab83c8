"""Seeded synthetic inputs of the shapes the head sees in training and inference.

Predictions: ``randn(batch, 5*Bx + C, S, S) * 2`` raw logits (the head tensor of
model.py:146-148, no activation applied).  Ground truth: per image 1..5 boxes,
centre uniform over the image, width/height uniform in [16, 300], clipped to
[0, img_size - 1] the way the reference's ``_crop_center`` clips (data.py:144),
label uniform over the classes; encoded by ``yolo_amd.targets``.
"""
from __future__ import annotations

import torch

from . import targets


def make_pred(batch: int, n_cells: int = 7, n_bboxes: int = 2, n_classes: int = 20,
              seed: int = 0, scale: float = 2.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    d = 5 * n_bboxes + n_classes
    return torch.randn(batch, d, n_cells, n_cells, generator=g, dtype=torch.float32) * scale


def make_gt_boxes(batch: int, n_classes: int = 20, img_size: int = 448, max_boxes: int = 5,
                  seed: int = 1):
    """Padded ground truth: ltrb (B, M, 4) f32, label (B, M) int64, count (B,) int64."""
    g = torch.Generator().manual_seed(seed)
    count = torch.randint(1, max_boxes + 1, (batch,), generator=g)
    cx = torch.rand(batch, max_boxes, generator=g) * img_size
    cy = torch.rand(batch, max_boxes, generator=g) * img_size
    w = 16.0 + torch.rand(batch, max_boxes, generator=g) * (300.0 - 16.0)
    h = 16.0 + torch.rand(batch, max_boxes, generator=g) * (300.0 - 16.0)
    ltrb = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], dim=-1)
    ltrb = torch.clip(ltrb, min=0, max=img_size - 1)
    label = torch.randint(0, n_classes, (batch, max_boxes), generator=g)
    return ltrb, label, count


def make_inputs(batch: int, n_cells: int = 7, n_bboxes: int = 2, n_classes: int = 20,
                img_size: int = 448, seed: int = 0):
    """pred, gt_norm_xywh, gt_cls_prob, obj_mask -- all CPU tensors."""
    pred = make_pred(batch, n_cells, n_bboxes, n_classes, seed=seed)
    ltrb, label, count = make_gt_boxes(batch, n_classes, img_size, seed=seed + 1)
    gx, gc, om = targets.encode_padded(ltrb, label, count, n_cells, n_classes, img_size)
    return pred, gx, gc, om
