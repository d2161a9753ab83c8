"""The multi-GPU branch of bench.py on the real backend.  A box has one GPU, so the
process group is RCCL ("nccl") at world size ONE (`--force-dist`): the run takes every
branch the N > 1 run takes -- init_process_group with the device, barriers, graph
replays, the in-place asynchronous all-reduce of each chunk's rows of the loss ring,
results(), the MAX all-reduce of the elapsed time, the all-gather of the per-rank
clocks, destroy_process_group -- and its logged loss rows must be the bits of the same
run without a process group (model.py:317's division by the batch, train.py:84-94)."""
import os

import numpy as np
import pytest

from test_bench_gpu import run_bench

pytestmark = pytest.mark.gpu


def test_collective_path_on_rccl_at_world_size_one(dev, tmp_path):
    common = ["--steps", "300", "--warmup", "8", "--measure", "step", "--prewarm-seconds", "0.05"]
    a, b = str(tmp_path / "dist.npy"), str(tmp_path / "plain.npy")
    line = run_bench(common + ["--force-dist", "--dump-log", a], drop=("MASTER_PORT", "MASTER_ADDR"))
    plain = run_bench(common + ["--dump-log", b])
    cfg = line["config"]
    assert cfg["backend"] == "nccl" and cfg["world_size"] == 1
    assert plain["config"]["backend"] is None
    assert line["n_gpus"] == 1 and line["steps"] == 300
    ranks = line["ranks"]
    assert ranks["ms_per_step_min"] > 0 and ranks["ms_per_step_max"] >= ranks["ms_per_step_min"]
    assert ranks["results_wait_ms_max"] >= 0
    rows_dist, rows_plain = np.load(a), np.load(b)
    assert rows_dist.shape == (300, 4)                 # three chunks: 128 + 128 + 44 steps
    assert np.array_equal(rows_dist.view(np.uint32), rows_plain.view(np.uint32))
    assert abs(line["loss"] - plain["loss"]) == 0.0
