"""bench.py's own launcher: `python bench.py --gpus N` must run N ranks or fail, never report a silent one-rank line.
CPU only: the `--stub` workload exercises the launcher, the rank protocol and the final all_gather over gloo."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)


def test_gpus_2_self_launches_two_ranks_gloo():
    r = run(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "1", "--batch", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["data"] == "stub"
    assert line["value"] > 0 and abs(line["value"] - 4 * 2 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]


def test_rank_count_mismatch_is_an_error():
    r = run(["--gpus", "2", "--stub"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_launcher_relays_a_failing_child():
    # no GPU here: the real workload's ranks fail loudly (no CPU fallback) and the launcher returns their non-zero code without a JSON line
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a GPU-less host")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0 and not r.stdout.strip()
