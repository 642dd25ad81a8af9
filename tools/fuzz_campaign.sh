#!/bin/bash
# One fuzz campaign on the tree as it stands (GPU box).  bash tools/fuzz_campaign.sh TAG FIRST
# Legs: subframe sweep with all signal kinds, VBS piece sizes with full order ranges, frame sweep, host-stream sweep, device-resident VBS sweep,
# long blocks.  Logs: gpurun_out/fuzz_TAG_*.log; the one-line totals: gpurun_out/fuzz_TAG_totals.txt
set -u
TAG=${1:-r03}; FIRST=${2:-3000000}
out=gpurun_out; mkdir -p $out
python -c "from flake_amd import srcid; print('kernel sources', srcid.kernel_sources_sha1())" > $out/fuzz_${TAG}_totals.txt 2>&1
run() {  # name, env..., -- pytest args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 1000 python -m pytest "$@" -q -m gpu -p no:cacheprovider > $out/fuzz_${TAG}_${name}.log 2>&1
    local rc=$?
    echo "$name: ${envs[*]} :: $(tail -1 $out/fuzz_${TAG}_${name}.log)" >> $out/fuzz_${TAG}_totals.txt
    [ $rc -le 1 ]          # a failed test goes on to the next leg; a timeout or crash stops the campaign
}
run sub FLAKE_FUZZ_FIRST=$FIRST FLAKE_FUZZ_SEEDS=${SUB:-24000} FLAKE_FUZZ_KINDS=1 FLAKE_FUZZ_FRAMES=1 FLAKE_FUZZ_FRAME_SEEDS=${FRM:-3000} FLAKE_FUZZ_PIECE_SEEDS=${PIECE:-6000} -- tests/test_gpu_fuzz.py &&
run host FLAKE_FUZZ_HOST_SEEDS=${HOST:-600} -- tests/test_host_frames.py &&
run vbs FLAKE_FUZZ_FIRST=$FIRST FLAKE_FUZZ_VBS_SEEDS=${VBS:-600} -- tests/test_gpu_vbs_dev.py &&
run long FLAKE_FUZZ_FIRST=$FIRST FLAKE_FUZZ_SEEDS=${LONGN:-600} FLAKE_FUZZ_LONG=1 FLAKE_FUZZ_FRAME_SEEDS=8 FLAKE_FUZZ_PIECE_SEEDS=8 -- tests/test_gpu_fuzz.py
cat $out/fuzz_${TAG}_totals.txt
