set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputests.log 2>&1; tail -3 gpurun_out/r02_gputests.log
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo bench rc $?
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- python3 bench.py --no-cpu-baseline > gpurun_out/r02_stats.log 2>&1; echo stats rc $?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_fetch -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --profile-steps 0 --no-cpu-baseline --no-other-configs > gpurun_out/r02_fetch.log 2>&1; echo fetch rc $?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_write -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --profile-steps 0 --no-cpu-baseline --no-other-configs > gpurun_out/r02_write.log 2>&1; echo write rc $?
python bench.py --gpus 2 --steps 100 --warmup 30 > gpurun_out/r02_bench_g2.json 2> gpurun_out/r02_bench_g2.err; echo g2 rc $?
