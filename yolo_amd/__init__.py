"""yolo_amd -- the YOLOv1 detection head (loss forward+backward, grid decode,
confidence threshold + per-class NMS) as hand-written HIP kernels for MI355X
(gfx950), behind the C ABI of include/yolo_head.h.

Only the head lives here.  ``YOLOv1Head`` mirrors the method names of the
reference's ``YOLOv1`` (model.py:95-434) for that path; ``targets`` and ``synth``
are host-side helpers that build its inputs.
"""
from . import _lib, synth, targets  # noqa: F401
from . import box_ops  # noqa: F401
from ._lib import (YH_MODE_DENSE, YH_MODE_REFERENCE, YoloHeadError)  # noqa: F401
from .head import LossLog, YOLOv1Head  # noqa: F401

MODE_REFERENCE = YH_MODE_REFERENCE
MODE_DENSE = YH_MODE_DENSE

__all__ = ["YOLOv1Head", "LossLog", "YoloHeadError", "MODE_REFERENCE", "MODE_DENSE", "targets", "synth", "box_ops"]
