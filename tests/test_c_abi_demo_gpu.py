"""The C ABI from plain C (examples/c_abi_demo.c, built with gcc: no PyTorch, no C++
in the caller) against the Python binding on the same inputs."""
import os
import subprocess

import numpy as np
import pytest
import torch

from yolo_amd import YOLOv1Head

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "build", "c_abi_demo")


def lcg_stream(n, state=12345):
    """The demo's generator: x <- 1664525 x + 1013904223 (mod 2^32), value (x >> 8) / 2^24."""
    out = np.empty(n, dtype=np.float32)
    x = state
    for i in range(n):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = np.float32(x >> 8) * np.float32(1.0 / 16777216.0)
    return out, x


@pytest.mark.skipif(not os.path.exists(DEMO), reason="build/c_abi_demo not built (make -C yolo_amd/csrc demo)")
def test_c_program_matches_python_binding(dev):
    B, S2, D, C = 8, 49, 30, 20
    res = subprocess.run([DEMO, str(B)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    tok = res.stdout.split()
    got = {tok[i]: float(tok[i + 1]) for i in range(0, len(tok), 2)}
    # the same inputs, drawn in the same order
    u, state = lcg_stream(B * D * S2)
    pred = (np.float32(6.0) * u - np.float32(3.0)).reshape(B, D, 7, 7)
    gx = np.zeros((B * S2, 4), np.float32)
    gc = np.zeros((B * S2, C), np.float32)
    om = np.zeros(B * S2, bool)
    for c in range(B * S2):
        v, state = lcg_stream(1, state)
        if v[0] < np.float32(0.0625):
            om[c] = True
            q, state = lcg_stream(4, state)
            gx[c] = np.float32(0.05) + np.float32(0.9) * q
            k, state = lcg_stream(1, state)
            gc[c, int(np.float32(k[0]) * np.float32(C))] = 1.0
    head = YOLOv1Head()
    p = torch.from_numpy(pred).to(dev)
    terms, grad = head.loss_fwd_bwd(p, torch.from_numpy(gx).view(B, S2, 1, 4).to(dev),
                                    torch.from_numpy(gc).view(B, S2, 1, C).to(dev),
                                    torch.from_numpy(om).view(B, S2, 1, 1).to(dev))
    _, count = head.postprocess(p, 0.5, 0.5)
    terms = terms.cpu().numpy()
    for name, want in (("loss", terms[4]), ("coord", terms[0]), ("conf", terms[1]), ("cls", terms[2])):
        assert abs(got[name] - float(want)) <= 1e-6 * abs(float(want)), (name, got[name], want)
    assert int(got["kept"]) == int(count.sum())
    gsum = float(grad.double().abs().sum())
    assert abs(got["gradsum"] - gsum) <= 1e-6 * gsum
