"""bench.py --gpus N with no launcher around it starts its own ranks (one child process per
GPU) and passes rank 0's JSON line on; the parent touches neither torch nor the GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


@pytest.mark.gpu
def test_two_ranks_from_a_bare_command_line():
    """`python bench.py --gpus 2` (gloo ranks sharing the one GPU): exit status 0 and exactly one
    JSON line, which names the world size the communicator reports and both ranks' launches."""
    r = _run(["--gpus", "2", "--size", "2049", "--steps", "3", "--warmup", "1"], {"MGX_DIST_BACKEND": "gloo"}, 600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    c = d["comm"]
    assert c["rccl_world"] == 2 and c["backend"] == "gloo" and c["launcher"] == "self"
    assert [i["rank"] for i in c["per_rank"]] == [0, 1]
    assert c["exchanges_per_cycle"] >= 1 and c["gathers_per_cycle"] >= 1
    assert d["resnorm_first_last"][1] < d["resnorm_first_last"][0]


@pytest.mark.gpu
def test_one_gpu_line_is_the_plain_path():
    """--gpus 1 starts no child: the line has no `comm` object and carries the level breakdown."""
    r = _run(["--gpus", "1", "--size", "2049", "--steps", "3", "--warmup", "1", "--cpu-baseline", "off"], {}, 600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip())
    assert d["n_gpus"] == 1 and "comm" not in d
    assert d["levels"][0]["level"] == 0 and d["levels"][0]["us_per_cycle"] > 0


def test_a_failing_rank_ends_the_launch_with_its_status():
    """Without a GPU every rank fails at start-up: the launcher must come back with a non-zero
    status (not hang in a collective, not print a line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU: the ranks would run")
    r = _run(["--gpus", "2", "--size", "257", "--steps", "1", "--warmup", "0"],
             {"MGX_DIST_BACKEND": "gloo", "MGX_BENCH_TIMEOUT": "120"}, 300)
    assert r.returncode != 0
    assert r.stdout.decode().strip() == ""
    assert b"rank 0 ended with status" in r.stderr or b"rank 1 ended with status" in r.stderr


@pytest.mark.gpu
def test_under_torch_distributed_run():
    """The way a multi-GPU scaling run is launched: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`: the script
    takes the ranks it is given (it starts none of its own) and rank 0 prints the one line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MGX_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--size", "2049", "--steps", "3", "--warmup", "1"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["comm"]["rccl_world"] == 2 and d["comm"]["launcher"] == "torch.distributed.run"
