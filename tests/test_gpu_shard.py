"""The N > 1 path with the HIP encoder on every rank.

Two ranks share cuda:0 (a one-GPU box; RCCL refuses two ranks on one card, so the
counters travel over gloo -- the data path has no collective at all).  Each rank
runs flake_amd.Encoder on its shard_range of the job with the job-wide frame
numbers; the concatenation of the ranks' frames must equal the single-rank
stream byte for byte, the reduced counters the totals.  A second test starts
`python bench.py --gpus 2` the way the driver does (no launcher) and reads the
line rank 0 prints."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import flake_amd
from flake_amd.shard import shard_range

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TOTAL, N = 37, 4096

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
import flake_amd
from flake_amd.shard import gather_frame_sizes, reduce_job_stats, shard_range

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
p = flake_amd.level_params(5)
first, last = shard_range({total}, rank, world)
pcm = flake_amd.synth_pcm(last - first, {n}, 2, 16, first_frame=first)
with flake_amd.Encoder(p, max_frames=last - first, device=0) as enc:      # the HIP path
    got = enc.encode_subframes(pcm, {n}, want_residual=False, want_frames=True,
                               first_frame_number=first)
sizes = [int(b) for b in got["frame_bytes"]]
frames = np.concatenate([got["frames"][f, :sizes[f]] for f in range(last - first)])
bits = int(got["info"]["rice_nbits"].clip(min=0).sum())
tot_frames, tot_bits, max_bytes = reduce_job_stats(last - first, bits, max(sizes))
all_sizes = gather_frame_sizes(sizes)
offset = int(sum(int(s.sum()) for s in all_sizes[:rank]))
np.save(os.path.join({out!r}, f"r{{rank}}.npy"), frames)
np.save(os.path.join({out!r}, f"m{{rank}}.npy"),
        np.array([tot_frames, tot_bits, max_bytes, offset, sum(sizes)], np.int64))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_hip_ranks_concatenate_to_single_rank_stream(tmp_path, decoder):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, total=TOTAL, n=N, out=str(tmp_path)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for pr in procs:
        assert pr.wait(timeout=600) == 0

    p = flake_amd.level_params(5)
    pcm = flake_amd.synth_pcm(TOTAL, N, 2, 16)
    with flake_amd.Encoder(p, max_frames=TOTAL) as enc:
        one = enc.encode_subframes(pcm, N, want_residual=False, want_frames=True)
    sizes = [int(b) for b in one["frame_bytes"]]
    single = np.concatenate([one["frames"][f, :sizes[f]] for f in range(TOTAL)])
    bits = int(one["info"]["rice_nbits"].clip(min=0).sum())
    parts = [np.load(tmp_path / f"r{r}.npy") for r in range(2)]
    metas = [np.load(tmp_path / f"m{r}.npy") for r in range(2)]
    assert [len(x) for x in parts] == [sum(sizes[slice(*shard_range(TOTAL, r, 2))]) for r in range(2)]
    assert (np.concatenate(parts) == single).all()
    for r, m in enumerate(metas):
        assert m[0] == TOTAL and m[1] == bits and m[2] == max(sizes)
        assert (single[m[3]:m[3] + m[4]] == parts[r]).all()       # placed by the gathered prefix
    out, fs = decoder.decode(np.concatenate(parts), 2, 16, TOTAL * N)
    assert len(fs) == TOTAL and (out.reshape(pcm.shape) == pcm).all()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as the driver calls it: the parent starts two child
    ranks (sharing this box's GPU, exchange over gloo) and rank 0's line comes back."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--steps", "5", "--warmup", "2", "--settle-ms", "1", "--frames", "256",
                        "--profile-steps", "0", "--no-cpu-baseline", "--no-other-configs"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # nothing but the JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["rank_frames"] == [256 * 5, 256 * 5] and out["job_frames"] == 2 * 256 * 5
    assert out["config"]["samples_per_step"] == 2 * 256 * 4096 * 2
    assert out["value"] > 0 and out["scaling"] == "weak"


def test_bench_strong_scaling_two_ranks():
    """`--scaling strong`: ONE job of --frames frames cut into contiguous shards
    (flake_amd.shard.shard_range), here 301 frames over two ranks: 151 + 150."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong",
                        "--steps", "5", "--warmup", "2", "--settle-ms", "1", "--frames", "301",
                        "--profile-steps", "0", "--no-cpu-baseline", "--no-other-configs"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["config"]["frames_per_gpu"] == [151, 150] == [b - a for a, b in (shard_range(301, 0, 2), shard_range(301, 1, 2))]
    assert out["rank_frames"] == [151 * 5, 150 * 5] and out["job_frames"] == 301 * 5
    assert out["config"]["samples_per_step"] == 301 * 4096 * 2
    assert out["job_residual_bits"] > 0 and out["value"] > 0


def test_bench_rccl_path_with_one_rank():
    """The collective path of bench.py over RCCL (init with device_id, all-reduce, barrier,
    all-gather on this rank's stream) -- what a one-GPU box can rehearse of the N-GPU run --
    and stdout carrying nothing but the one JSON line (RCCL prints a banner on stdout)."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", BENCH_DIST_BACKEND="nccl",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--steps", "5", "--warmup", "2", "--settle-ms", "1", "--frames", "256",
                        "--profile-steps", "0", "--no-cpu-baseline", "--no-other-configs"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["dist_backend"] == "nccl" and out["ranks_seen"] == 1 and out["n_gpus"] == 1
    assert out["job_frames"] == 256 * 5
