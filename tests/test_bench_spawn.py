"""`python bench.py --gpus N` with no launcher environment must start N ranks itself (never run one rank and report N):
the self-launch path of bench.py, exercised on CPU with the gloo backend through `--spawn-check` (process group up, one
all-reduce, the line's `n_gpus` = the world size seen after init)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    r = _run(["--gpus", "2", "--spawn-check"], {"SGR_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_summed"] == 2 and d["backend"] == "gloo"


def test_more_ranks_than_gpus_is_refused_under_rccl():
    # (no GPU in the build container: any N > 1 exceeds the device count; on a one-GPU box the same holds for N = 2)
    r = _run(["--gpus", "64", "--spawn-check"], {"SGR_BENCH_BACKEND": "nccl"})
    assert r.returncode == 2
    assert b"RCCL needs one GPU per rank" in r.stderr


def test_a_launcher_world_size_that_contradicts_gpus_is_refused():
    r = _run(["--gpus", "2", "--spawn-check"], {"SGR_BENCH_BACKEND": "gloo", "WORLD_SIZE": "1", "RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                              "MASTER_PORT": "29577"})
    assert r.returncode != 0 and b"refusing" in r.stderr
