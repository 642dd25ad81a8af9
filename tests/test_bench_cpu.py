"""bench.py on a machine without a GPU: the parts that do not need one.  `--gpus N` (N > 1)
without a launcher must start N child ranks (torch.distributed.run) from a parent that never
touches the GPU, and a rank without a GPU must fail loudly -- there is no CPU fallback."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="this is the no-GPU behaviour")
def test_self_launch_reaches_the_ranks_and_fails_loudly_without_a_gpu():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-1500:]     # both child ranks said so
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]       # and no result line


def test_host_cores_respects_the_cap(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("BENCH_CPU_CORES", "3")
    used, avail = bench.host_cores()
    assert 1 <= used <= 3 and used <= avail <= len(os.sched_getaffinity(0))
    monkeypatch.setenv("BENCH_CPU_CORES", "100000")
    used, avail = bench.host_cores()
    assert 1 <= used == avail <= len(os.sched_getaffinity(0))
