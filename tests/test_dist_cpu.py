"""The N > 1 path on the CPU: shard arithmetic and the batched loss all-reduce
under gloo with two ranks (no GPU, no HIP library calls)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu
from yolo_amd import synth
from yolo_amd.dist import LossAllReducer, shard_range


def test_shard_range_partitions_the_batch():
    for gb in (0, 1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(65536, 3, 8) == (24576, 32768)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_single_process_reducer_is_a_plain_log():
    red = LossAllReducer()
    ring = torch.arange(32, dtype=torch.float32).view(8, 4)
    red.flush(ring, 8)
    ring += 100
    red.flush(ring, 3)
    red.flush(ring, 0)
    out = red.results()
    assert out.shape == (11, 4)
    assert torch.equal(out[:8], torch.arange(32, dtype=torch.float32).view(8, 4))
    assert torch.equal(out[8:], ring[:3])
    with pytest.raises(ValueError):
        red.flush(ring, 9)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, global_batch, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        lo, hi = shard_range(global_batch, rank, world)
        red = LossAllReducer()
        ring = torch.zeros(4, 4)
        rows = 0
        for s in range(steps):
            # what one rank's loss kernel leaves in its ring: the SUMS over its shard
            pred, gx, gc, om = synth.make_inputs(global_batch, seed=100 + s)
            c, f, k, _ = ref_cpu.loss_terms(pred[lo:hi], gx[lo:hi], gc[lo:hi], om[lo:hi])
            ring[rows] = torch.stack([c, f, k, c + f + k])
            rows += 1
            if rows == ring.shape[0]:
                red.flush(ring, rows)
                rows = 0
        red.flush(ring, rows)
        torch.save(red.results(), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_loss_all_reduce_matches_the_unsharded_loss(tmp_path):
    world, global_batch, steps = 2, 13, 6          # ragged shards: 7 + 6 images
    mp.spawn(_worker, args=(world, _free_port(), global_batch, steps, str(tmp_path)),
             nprocs=world, join=True)
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    assert torch.equal(got[0], got[1]) and got[0].shape == (steps, 4)
    for s in range(steps):
        pred, gx, gc, om = synth.make_inputs(global_batch, seed=100 + s)
        want = ref_cpu.get_loss(pred, gx, gc, om).item()
        assert abs(got[0][s, 3].item() / global_batch - want) <= 1e-5 * abs(want)


def test_shard_gradients_are_slices_of_the_global_gradient():
    """With global_batch passed down, a rank's gradient needs no exchange at all."""
    pred, gx, gc, om = synth.make_inputs(13, seed=7)
    full = ref_cpu.loss_and_grad(pred, gx, gc, om)
    for rank in range(2):
        lo, hi = shard_range(13, rank, 2)
        part = ref_cpu.loss_and_grad(pred[lo:hi], gx[lo:hi], gc[lo:hi], om[lo:hi], global_batch=13)
        torch.testing.assert_close(part["grad"], full["grad"][lo:hi], rtol=1e-6, atol=1e-9)
