"""bench.py as its own multi-rank launcher: the command it builds, and a 2-rank gloo rehearsal without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launch_command():
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], 29511, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5"]      # the ranks see the caller's own arguments


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "n_gpus must be the world size" in r.stderr


def test_two_rank_gloo_dry_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--backend", "gloo",
                        "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["dry"] is True
    assert out["loss"] == 4.5                                    # last of the 4 steps: mean over the two ranks of 1 + rank + 3


def test_strong_case_flags_go_together():
    import bench
    assert bench.main(["--case", "gray_scott"]) == 2
    assert bench.main(["--strong"]) == 2
    assert bench.parse_args(["--case", "gray_scott", "--strong", "--gpus", "4"]).strong
