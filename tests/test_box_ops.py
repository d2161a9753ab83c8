"""Pairwise box geometry (eval.py:22-83): oracle against the reference fixture on
the CPU; the HIP kernels bit-exact against the same fixture on the GPU."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_oracle_giou_matches_reference(golden):
    g = golden("g1_iou.npz")
    giou = ref_cpu.get_giou(t(g["ltrb1"]), t(g["ltrb2"]))
    assert torch.equal(giou, t(g["giou"]))
    assert giou[5, 5] == -1            # both boxes empty at the same point: enclosing area 0
    assert giou[2, 2] == 1             # identical boxes


def test_oracle_enclosing_area_is_the_one_inside_the_reference_giou(golden):
    """eval.py:68-73 has no fixture of its own; the reference's GIoU (fixture g1) is built from
    it, and rebuilding that GIoU from the oracle's enclosing area gives the fixture's bits."""
    g = golden("g1_iou.npz")
    b1, b2 = t(g["ltrb1"]), t(g["ltrb2"])
    c = ref_cpu.get_smallest_enclosing_area(b1, b2)
    union = ref_cpu.get_area(b1)[:, None] + ref_cpu.get_area(b2)[None, :] - ref_cpu.get_intersection_area(b1, b2)
    giou = torch.where(c == 0, -1, ref_cpu.get_iou(b1, b2) - ((c - union) / c))
    assert torch.equal(giou, t(g["giou"]))


@pytest.mark.gpu
def test_gpu_geometry_bit_exact_against_reference_fixture(dev, golden):
    from yolo_amd import box_ops

    g = golden("g1_iou.npz")
    b1, b2 = t(g["ltrb1"]).to(dev), t(g["ltrb2"]).to(dev)
    for fn, key in ((box_ops.get_iou, "iou"), (box_ops.get_giou, "giou"),
                    (box_ops.get_intersection_area, "inter")):
        got = fn(b1, b2).cpu().numpy()
        np.testing.assert_array_equal(got.view(np.uint32), g[key].view(np.uint32))
    enc = box_ops.get_smallest_enclosing_area(b1, b2).cpu()
    assert torch.equal(enc, ref_cpu.get_smallest_enclosing_area(b1.cpu(), b2.cpu()))
    np.testing.assert_array_equal(box_ops.get_area(b1).cpu().numpy(), g["area1"])
    np.testing.assert_array_equal(box_ops.get_area(b2).cpu().numpy(), g["area2"])


@pytest.mark.gpu
def test_gpu_geometry_larger_and_ragged(dev):
    from yolo_amd import box_ops

    gen = torch.Generator().manual_seed(9)
    for n, m in ((1, 1), (32, 1024), (1000, 37), (0, 5), (3, 0)):   # (32, 1024): eval.py:152-173
        a = torch.rand(n, 4, generator=gen) * 64
        b = torch.rand(m, 4, generator=gen) * 64                    # many inverted boxes
        for fn, ref in ((box_ops.get_iou, ref_cpu.get_iou), (box_ops.get_giou, ref_cpu.get_giou),
                        (box_ops.get_intersection_area, ref_cpu.get_intersection_area),
                        (box_ops.get_smallest_enclosing_area, ref_cpu.get_smallest_enclosing_area)):
            got = fn(a.to(dev), b.to(dev)).cpu()
            want = ref(a, b)
            assert got.shape == want.shape
            assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    with pytest.raises(ValueError):
        box_ops.get_iou(torch.zeros(3, 5, device=dev), torch.zeros(3, 4, device=dev))
