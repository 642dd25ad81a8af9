# One round of evidence on a gpurun box: GPU tests, bench, rocprofv3 kernel stats, the PMC passes
# (separate runs, --kernel-trace only beside --pmc), a two-rank rehearsal.  usage: bash tools/prof_round.sh TAG
set -o pipefail
T=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gputests.log 2>&1; tail -3 gpurun_out/${T}_gputests.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo bench rc $?
# the headline alone (its per-kernel averages must agree with roofline.kernel_ms), then everything else
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --no-cpu-baseline --no-other-configs > gpurun_out/${T}_stats.log 2>&1; echo stats rc $?
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats_other -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --profile-steps 0 > gpurun_out/${T}_stats_other.log 2>&1; echo stats other rc $?
P="--steps 3 --warmup 1 --settle-ms 0 --profile-steps 0 --no-cpu-baseline --no-other-configs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_fetch -- python3 bench.py $P > gpurun_out/${T}_fetch.log 2>&1; echo fetch rc $?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_write -- python3 bench.py $P > gpurun_out/${T}_write.log 2>&1; echo write rc $?
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${T}_sq -- python3 bench.py $P > gpurun_out/${T}_sq.log 2>&1; echo sq rc $?
python bench.py --gpus 2 --steps 100 --warmup 30 > gpurun_out/${T}_bench_g2.json 2> gpurun_out/${T}_bench_g2.err; echo g2 rc $?
python bench.py --gpus 2 --scaling strong --steps 100 --warmup 30 --no-other-configs > gpurun_out/${T}_bench_g2_strong.json 2> gpurun_out/${T}_bench_g2_strong.err; echo g2 strong rc $?
