"""Pairwise box geometry on the GPU with the names and shapes of the reference's
eval.py: get_area (eval.py:22-32), get_intersection_area (:35-50), get_iou
(:53-65), get_giou (:68-83).  fp32 results are bit-identical to the reference's
PyTorch expressions."""
from __future__ import annotations

import torch

from . import _lib
from .head import _need, _stream


def _boxes(t: torch.Tensor, name: str) -> torch.Tensor:
    _need(t, name, torch.float32)
    if t.dim() != 2 or t.shape[1] != 4:
        raise ValueError(f"{name}: expected shape (N, 4), got {tuple(t.shape)}")
    return t


def _pairwise(ltrb1, ltrb2, kind):
    _boxes(ltrb1, "ltrb1")
    _boxes(ltrb2, "ltrb2")
    if ltrb1.device != ltrb2.device:
        raise ValueError("ltrb1 and ltrb2 live on different devices")
    n, m = ltrb1.shape[0], ltrb2.shape[0]
    out = torch.empty(n, m, dtype=torch.float32, device=ltrb1.device)
    with torch.cuda.device(ltrb1.device):
        _lib.check(_lib.load().yh_pairwise_boxes(ltrb1.data_ptr(), ltrb2.data_ptr(), out.data_ptr(),
                                                 n, m, kind, _stream(ltrb1.device)))
    return out


def get_area(ltrb):
    """eval.py:22-32: (N, 4) -> (N)."""
    _boxes(ltrb, "ltrb")
    out = torch.empty(ltrb.shape[0], dtype=torch.float32, device=ltrb.device)
    with torch.cuda.device(ltrb.device):
        _lib.check(_lib.load().yh_box_area(ltrb.data_ptr(), out.data_ptr(), ltrb.shape[0],
                                           _stream(ltrb.device)))
    return out


def get_intersection_area(ltrb1, ltrb2):
    """eval.py:35-50: (N, 4), (M, 4) -> (N, M)."""
    return _pairwise(ltrb1, ltrb2, _lib.YH_BOX_INTERSECTION)


def get_iou(ltrb1, ltrb2):
    """eval.py:53-65: (N, 4), (M, 4) -> (N, M); 0/0 -> 0."""
    return _pairwise(ltrb1, ltrb2, _lib.YH_BOX_IOU)


def get_giou(bbox1, bbox2):
    """eval.py:76-83: (N, 4), (M, 4) -> (N, M); -1 where the enclosing box is empty."""
    return _pairwise(bbox1, bbox2, _lib.YH_BOX_GIOU)
