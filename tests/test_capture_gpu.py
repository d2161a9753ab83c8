"""The entry points that sort in global memory -- yh_nms_ws above 2,048 candidates per image
and yh_map_11point above 1,024 predictions (eval.py:121's ranking) -- only enqueue kernels:
captured into a HIP graph and replayed they give the bits of the eager call, which the other
tests hold against the oracle."""
import pytest
import torch

from yolo_amd import _lib

pytestmark = pytest.mark.gpu


def test_large_nms_under_graph_capture(dev):
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    batch, n = 3, 5000
    c = torch.rand(batch, n, 2, generator=g) * 448
    wh = torch.rand(batch, n, 2, generator=g) * 120 + 4
    ltrb = torch.cat([(c - wh / 2).clamp(min=0), (c + wh / 2).clamp(max=448)], dim=-1).contiguous().to(dev)
    score = torch.rand(batch, n, generator=g).to(dev)
    score[0, 10] = score[0, 4000]                      # equal scores: index order decides
    cls = torch.randint(0, 300, (batch, n), generator=g, dtype=torch.int32).to(dev)
    ws = torch.empty(lib.yh_nms_workspace_bytes(batch, n, 0), dtype=torch.uint8, device=dev)

    def call(dets, cnt, kidx, stream):
        _lib.check(lib.yh_nms_ws(ltrb.data_ptr(), score.data_ptr(), cls.data_ptr(), None, dets.data_ptr(),
                                 cnt.data_ptr(), kidx.data_ptr(), 0.2, 0.5, batch, n, ws.data_ptr(), ws.numel(),
                                 stream.cuda_stream))

    def outputs():
        return (torch.full((batch, n, 6), float("nan"), device=dev),
                torch.full((batch,), -7, dtype=torch.int32, device=dev),
                torch.full((batch, n), -7, dtype=torch.int32, device=dev))

    want = outputs()
    call(*want, torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    got = outputs()
    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        call(*got, side)
    for _ in range(2):
        for t in got:
            t.fill_(-3)
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert int(want[1].min()) > 100


def test_map_under_graph_capture(dev):
    lib = _lib.load()
    g = torch.Generator().manual_seed(6)
    P, G, ncls = 30000, 3000, 20
    gb = torch.rand(G, 2, generator=g) * 400
    gt = torch.cat([gb, gb + 10 + torch.rand(G, 2, generator=g) * 40,
                    torch.randint(0, ncls, (G, 1), generator=g).float()], dim=1).to(dev)
    src = torch.randint(0, G, (P,), generator=g)
    pred = torch.cat([gt.cpu()[src, :4] + torch.randn(P, 4, generator=g) * 6, torch.rand(P, 1, generator=g),
                      gt.cpu()[src, 4:5]], dim=1).to(dev)
    ws = torch.empty(lib.yh_map_workspace_bytes(P, G), dtype=torch.uint8, device=dev)

    def call(ap, m, stream):
        _lib.check(lib.yh_map_11point(pred.data_ptr(), gt.data_ptr(), P, G, ncls, 4.0, ap.data_ptr(), m.data_ptr(),
                                      ws.data_ptr(), ws.numel(), stream.cuda_stream))

    want = (torch.zeros(ncls, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev))
    call(*want, torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    got = (torch.zeros_like(want[0]), torch.zeros_like(want[1]))
    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        call(*got, side)
    for _ in range(2):
        got[0].fill_(-1)
        got[1].fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert want[1].item() > 0.01
