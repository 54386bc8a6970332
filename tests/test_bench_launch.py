"""bench.py's own launcher (no GPU needed): `python bench.py --gpus N` with WORLD_SIZE unset must start its N ranks itself --
the driver types exactly that -- relay rank 0's single JSON line and exit non-zero when a rank fails."""
import json
import os
import subprocess
import sys

from conftest import REPO


def _bench(args, env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          env=e, timeout=600)


def test_bare_command_spawns_its_ranks_and_relays_one_line():
    p = _bench(["--gpus", "3", "--steps", "2", "--warmup", "1"], {"TD_BENCH_DRYRUN": "1"})
    assert p.returncode == 0, p.stderr.decode()[-1500:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dryrun"] and d["n_gpus"] == 3 and d["rank_sum"] == 6 and d["local_ranks_seen"] == "3"


def test_a_failing_rank_fails_the_command():
    """Without a GPU every rank stops at "no GPU visible" (there is no CPU fallback): the parent must exit non-zero and print no
    result line."""
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--extras", "0", "--cpu-sample", "0"], {"HIP_VISIBLE_DEVICES": "-1", "CUDA_VISIBLE_DEVICES": ""})
    assert p.returncode != 0
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert "no GPU visible" in p.stderr.decode()


def test_under_a_launcher_the_world_must_match():
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "TD_BENCH_DRYRUN": "1"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr.decode()
