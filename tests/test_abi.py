"""The C ABI without a GPU: the library loads, exports every symbol that
include/yolo_head.h declares, and rejects bad arguments on the host before any
kernel launch (so these calls are safe on a machine with no device)."""
import ctypes
import os
import re

import pytest

from yolo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "yolo_head.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yh_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert names, "no declarations parsed"
    assert names == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_version_and_strerror():
    lib = _lib.load()
    assert lib.yh_version() == 101
    assert _lib.strerror(0) == "ok"
    for rc in range(-7, 0):
        assert _lib.strerror(rc).startswith("yolo_head:")
    assert "unknown" in _lib.strerror(-99)


def test_constants_match_header():
    text = open(HEADER).read()
    for name in ("YH_ERR_NULL", "YH_ERR_SHAPE", "YH_ERR_WORKSPACE", "YH_ERR_ALIGN", "YH_ERR_ARG",
                 "YH_ERR_UNSUPPORTED", "YH_ERR_SCRATCH", "YH_STATIC_SCRATCH_BYTES", "YH_OP_LOSS", "YH_MODE_REFERENCE", "YH_MODE_DENSE",
                 "YH_MAX_CANDIDATES"):
        m = re.search(rf"#define {name} \(?(-?\d+)\)?", text)
        assert m and int(m.group(1)) == getattr(_lib, name), name


def test_workspace_bytes():
    lib = _lib.load()
    # S=7: four images per 256-thread workgroup, one 16-byte row per workgroup
    assert lib.yh_workspace_bytes(_lib.YH_OP_LOSS, 4096, 7, 2, 20) == (4096 // 4) * 16
    assert lib.yh_workspace_bytes(_lib.YH_OP_LOSS, 0, 7, 2, 20) == 16
    assert lib.yh_workspace_bytes(_lib.YH_OP_LOSS, 16, 14, 5, 80) == 16 * 16
    assert lib.yh_workspace_bytes(_lib.YH_OP_DECODE_NMS, 4096, 7, 2, 20) == 0


FAKE = 0x1000  # never dereferenced: every call below fails its host-side checks first


@pytest.mark.parametrize("kw,rc", [
    (dict(S=0), _lib.YH_ERR_SHAPE),
    (dict(S=65), _lib.YH_ERR_SHAPE),
    (dict(Bx=0), _lib.YH_ERR_SHAPE),
    (dict(Bx=17), _lib.YH_ERR_SHAPE),
    (dict(C=0), _lib.YH_ERR_SHAPE),
    (dict(batch=-1), _lib.YH_ERR_SHAPE),
    (dict(img=450), _lib.YH_ERR_SHAPE),        # 450 % 7 != 0
    (dict(pred=0), _lib.YH_ERR_NULL),
    (dict(terms=0), _lib.YH_ERR_NULL),
    (dict(ws=0), _lib.YH_ERR_WORKSPACE),
    (dict(ws_bytes=8), _lib.YH_ERR_WORKSPACE),
    (dict(gx=FAKE + 4), _lib.YH_ERR_ALIGN),
    (dict(lc=float("nan")), _lib.YH_ERR_ARG),
])
def test_loss_argument_errors(kw, rc):
    a = dict(pred=FAKE, gx=FAKE, gc=FAKE, om=FAKE, grad=FAKE, terms=FAKE, ws=FAKE,
             ws_bytes=1 << 20, batch=16, S=7, Bx=2, C=20, img=448, lc=5.0, ln=0.5, inv=1 / 16)
    a.update(kw)
    got = _lib.load().yh_loss_fwd_bwd(a["pred"], a["gx"], a["gc"], a["om"], a["grad"], a["terms"],
                                      a["ws"], a["ws_bytes"], a["batch"], a["S"], a["Bx"], a["C"],
                                      a["img"], a["lc"], a["ln"], a["inv"], None)
    assert got == rc


def test_nms_argument_errors():
    lib = _lib.load()
    f = FAKE
    assert lib.yh_decode_nms(f, f, f, None, 2, 0.5, 0.5, 4, 7, 2, 20, 448, None) == _lib.YH_ERR_ARG
    assert lib.yh_decode_nms(f, f, f, None, 0, float("nan"), 0.5, 4, 7, 2, 20, 448, None) == _lib.YH_ERR_ARG
    assert lib.yh_decode_nms(None, f, f, None, 0, 0.5, 0.5, 4, 7, 2, 20, 448, None) == _lib.YH_ERR_NULL
    # 32*32*2 = 2048 candidates per image
    assert lib.yh_decode_nms(f, f, f, None, 1, 0.5, 0.5, 4, 32, 3, 20, 448, None) == _lib.YH_ERR_UNSUPPORTED
    assert lib.yh_nms(f, f, f, None, f, f, None, 0.5, 0.5, 4, 2049, None) == _lib.YH_ERR_UNSUPPORTED
    assert lib.yh_nms(f, f, f, None, f, f, None, 0.5, 0.5, 4, 0, None) == _lib.YH_ERR_SHAPE
    assert lib.yh_nms(f + 4, f, f, None, f, f, None, 0.5, 0.5, 4, 10, None) == _lib.YH_ERR_ALIGN
    # batch 0 is a valid no-op everywhere
    assert lib.yh_decode_nms(None, None, None, None, 0, 0.5, 0.5, 0, 7, 2, 20, 448, None) == 0
    assert lib.yh_nms(None, None, None, None, None, None, None, 0.5, 0.5, 0, 10, None) == 0
    assert lib.yh_decode_grid(None, None, None, None, None, None, None, None, 0.5, 0, 7, 2, 20, 448, None) == 0
    assert lib.yh_decode_candidates(f, f, f, f, f, 3, 4, 7, 2, 20, 448, None) == _lib.YH_ERR_ARG
    assert lib.yh_scale_grad(None, None, 0, None) == 0
    assert lib.yh_scale_grad(None, f, 8, None) == _lib.YH_ERR_NULL


def test_head_refuses_cpu_tensors():
    import torch
    from yolo_amd import YOLOv1Head, synth

    head = YOLOv1Head()
    pred, gx, gc, om = synth.make_inputs(2)
    with pytest.raises(ValueError, match="GPU"):
        head.get_loss(pred, gx, gc, om)
    with pytest.raises(ValueError, match="GPU"):
        head.postprocess(pred)
    with pytest.raises(ValueError, match="shape"):
        head._check_out(torch.zeros(2, 31, 7, 7))


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libyolohead.so")
    with pytest.raises(ImportError, match="no CPU or PyTorch fallback"):
        _lib.load()
