"""`python bench.py --gpus N` must start itself for N > 1 (the driver invokes it the same way it does for N = 1): without a
launcher it re-execs under torch.distributed.run (127.0.0.1), rank 0 prints ONE JSON line.  Exercised here over gloo with
--dry (no model, no timed region: this container has no GPU); too few visible GPUs is an immediate, clear error."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=180):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_bench_gpus2_launches_itself_and_prints_one_line():
    r = _run(["--gpus", "2", "--dry", "--steps", "7", "--warmup", "3"], {"CAL_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["warmup"] == 3 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 256
    assert out["data_parallel"]["rccl_ranks_seen"] == 2
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert key in out


def test_bench_gpus_more_than_visible_fails_fast_with_a_message():
    r = _run(["--gpus", "2"], {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""}, timeout=120)
    assert r.returncode != 0
    assert "only 0 GPU(s) visible" in (r.stderr + r.stdout)


def test_bench_gpus_disagreeing_with_the_launcher_is_an_error():
    r = _run(["--gpus", "4", "--dry"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
