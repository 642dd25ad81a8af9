"""Device-resident rate of the VBS presets (BASELINE configs[4]): blocks resident in HBM ->
packed FLAC stream in HBM through fhip_encode_blocks_vbs_dev (no host sync inside).
python tools/vbs_dev_bench.py [blocks] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flake_amd

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
for level in [int(v) for v in os.environ.get("VBS_LEVELS", "9,10,12").split(",")]:
    p = flake_amd.level_params(level)
    n = p.block_size
    pcm = flake_amd.synth_pcm(nblk, n, 2, 16)
    pcm[::3, n // 2:, :] //= 16          # a transient in every third block: something to split
    d_pcm = torch.from_numpy(pcm).to(dev)
    cap = pcm.size * 5
    packed = torch.zeros(cap, dtype=torch.uint8, device=dev)
    totals = torch.zeros(4, dtype=torch.int64, device=dev)
    enc = flake_amd.Encoder(p, max_frames=8 * nblk)
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    enc.set_stream(st.cuda_stream)
    def step():
        enc.encode_blocks_vbs_dev(d_pcm, nblk, n, packed, cap, totals)
    for _ in range(3): step()
    enc.sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        step(); enc.sync()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    enc.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    enc.set_profiling(True); enc.kernel_times(reset=True)
    for _ in range(3): step()
    enc.sync()
    per = {k: round(t / 3, 4) for k, (t, c) in enc.kernel_times(reset=True).items() if c}
    enc.set_profiling(False)
    t = totals.cpu().numpy()
    print(f"level {level} n={n}: {nblk} blocks -> {t[0]} frames, {t[1]} bytes; {ms:.3f} ms/batch, "
          f"{nblk * n * 2 / ms / 1e3:.0f} Msamples/s; serial kernel sums (ms) {per}", flush=True)
    enc.close()
