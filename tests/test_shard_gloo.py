"""The N > 1 path on CPU: world_size-2 gloo.  Each rank encodes its contiguous
frame shard (the oracle stands in for the GPU here); the concatenation of the
shards must equal the single-process result byte for byte, and the reduced job
counters must equal the totals."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import flake_amd
from flake_amd.shard import gather_frame_sizes, reduce_job_stats, shard_range

HERE = os.path.dirname(os.path.abspath(__file__))
TOTAL, N = 11, 1024


def test_shard_range_partitions():
    for total in (0, 1, 7, 8, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, HERE)
    import oraclelib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = flake_amd.level_params(5)
    p.block_size = N
    first, last = shard_range(TOTAL, rank, world)
    pcm = flake_amd.synth_pcm(last - first, N, 2, 16, first_frame=first)
    orc = oraclelib.Oracle()
    frames, sizes, bits = [], [], 0
    for i in range(last - first):
        rc, fb, sf, _, _ = orc.encode_frame(p, first + i, pcm[i], N)
        assert rc > 0
        frames.append(fb)
        sizes.append(rc)
    out = orc.encode_subframes_batch(p, pcm, N, slot_bytes=0)
    bits = int(out["info"]["rice_nbits"].clip(min=0).sum())
    tot_frames, tot_bits, max_bytes = reduce_job_stats(last - first, bits, max(sizes))
    all_sizes = gather_frame_sizes(sizes)
    offset = int(sum(int(s.sum()) for s in all_sizes[:rank]))
    np.save(os.path.join(outdir, f"r{rank}.npy"), np.concatenate(frames))
    np.save(os.path.join(outdir, f"m{rank}.npy"),
            np.array([tot_frames, tot_bits, max_bytes, offset, sum(sizes)], np.int64))
    dist.destroy_process_group()


def test_two_rank_shards_concatenate_to_single_rank_output(tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p = flake_amd.level_params(5)
    p.block_size = N
    pcm = flake_amd.synth_pcm(TOTAL, N, 2, 16)
    single, sizes = [], []
    for i in range(TOTAL):
        rc, fb, _, _, _ = oracle.encode_frame(p, i, pcm[i], N)
        single.append(fb)
        sizes.append(rc)
    single = np.concatenate(single)
    bits = int(oracle.encode_subframes_batch(p, pcm, N, slot_bytes=0)["info"]["rice_nbits"].clip(min=0).sum())
    parts = [np.load(tmp_path / f"r{r}.npy") for r in range(2)]
    metas = [np.load(tmp_path / f"m{r}.npy") for r in range(2)]
    assert (np.concatenate(parts) == single).all()
    for r, m in enumerate(metas):
        assert m[0] == TOTAL and m[1] == bits and m[2] == max(sizes)
        assert (single[m[3]:m[3] + m[4]] == parts[r]).all()      # placed by the gathered prefix
