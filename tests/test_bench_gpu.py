"""bench.py as the driver starts it: `--gpus 2` as a plain command launches its own two
ranks (model.py:317's division by the batch made global, train.py:84-94's loop), here
both on GPU 0 over gloo; and the one-GPU line carries the keys the contract names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, extra_env=None, timeout=300, drop=()):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    for name in drop:
        env.pop(name, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True,
                         text=True, timeout=timeout, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_self_launched_on_config_3(dev):
    line = run_bench(["--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "4",
                      "--measure", "step", "--prewarm-seconds", "0.05"],
                     {"BENCH_SINGLE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["scaling"] == "strong"
    cfg = line["config"]
    assert cfg["global_batch"] == 65536 and cfg["images_per_gpu"] == 32768
    assert cfg["world_size"] == 2 and cfg["backend"] == "gloo"
    assert line["value"] > 0 and abs(line["value"] - 20 * 65536 / (line["ms_per_step"] * 20e-3)) < 1e-3 * line["value"]
    assert 20.0 < line["loss"] < 40.0          # the loss of the whole batch, divided by the global batch


def test_one_gpu_line_has_the_contract_keys(dev):
    line = run_bench(["--steps", "20", "--warmup", "4", "--kernel-iters", "40", "--no-stress",
                      "--cpu-seconds", "2", "--prewarm-seconds", "0.05"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / roof["kernel_us"] / 1e3) < 1e-6 * roof["achieved"]
    assert roof["regime"].startswith("buffers rotate")
    # the headline value is the HBM-proper figure: the timed region rotates through buffer sets
    assert line["config"]["regime"].startswith("buffers rotate") and line["value_hbm_proper"] == line["value"]
    assert line["cache_resident"]["images_per_s"] > 0
    assert roof["copy_ceiling"]["gbs"] > 3000 and roof["copy_ceiling"]["read_only_gbs"] > 3000
    assert roof["decode_nms"]["rows_written_per_image"] == 49
    rows = line["configs"]["dp_strong_n1"]["rows"]
    assert [r["images_on_this_gpu"] for r in rows] == [65536, 32768, 16384, 8192]
    assert all(r["step_us"] > 0 for r in rows)
    cb = line["cpu_baseline"]["detail"]["batch4096"]
    assert all(v["timed_passes"] == 20 and v["warmup_passes"] == 3 for v in cb.values())
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    assert line["config"]["workload"].startswith("YOLOv1 head S=7,B=2,C=20 batch 4096")


def test_other_ways_to_run_the_step(dev):
    """--step two-stream (the two kernels on two streams, as before the one-kernel step) and
    --regime resident (one buffer set, served by the memory-side cache in part) still run and
    say what they ran; the loss they log is the loss of the default run."""
    base = ["--steps", "40", "--warmup", "4", "--measure", "step", "--prewarm-seconds", "0.05"]
    fused = run_bench(base)
    two = run_bench(base + ["--step", "two-stream"])
    res = run_bench(base + ["--regime", "resident"])
    assert fused["config"]["streams"] == 1 and "yh_head_step_pipelined" in fused["config"]["step"]
    assert two["config"]["streams"] == 2 and "yh_loss_fwd_bwd_pipelined" in two["config"]["step"]
    assert res["config"]["regime"].startswith("one resident buffer set") and "value_hbm_proper" not in res
    assert fused["config"]["regime"].startswith("buffers rotate") and fused["value_hbm_proper"] == fused["value"]
    assert fused["loss"] == two["loss"] == res["loss"]
