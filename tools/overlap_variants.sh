cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in "plain" "ahead" "ahead FHIP_AHEAD_PRIO=1" "ahead FHIP_AHEAD_GATE=1" "ahead FHIP_AHEAD_PRIO=1 FHIP_AHEAD_GATE=1"; do
  set -- $V; MODE=$1; shift
  FLAGS=""; [ "$MODE" = "ahead" ] && FLAGS="--ahead"
  env "$@" python bench.py $FLAGS --no-cpu-baseline --no-other-configs --steps 400 --warmup 150 > gpurun_out/r04_ov.json 2> gpurun_out/r04_ov.err
  python - "$V" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04_ov.json')); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'])
PY
done
FHIP_AHEAD_PRIO=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_trace_prio -- python3 tools/sq_case.py ahead 40 > gpurun_out/r04_trace_prio.log 2>&1 && python3 tools/overlap_trace.py gpurun_out/r04_trace_prio 4 | tail -16
