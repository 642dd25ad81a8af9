# Round 4, what the search kernels (and the others of each config) spend: ONE rocprofv3 --kernel-trace --stats
# pass PER CONFIG (so a row's dominant-kernel average can be recomputed) and two SQ --pmc passes per config
# (--kernel-trace only beside --pmc; python3 directly after --).   bash tools/r04_measure.sh TAG "c2 l8 l10 l12 c3"
set -o pipefail
T=${1:-r04a}
CASES=${2:-"c2 l8 l10 l12 c3"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in $CASES; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats_${C} -- python3 tools/sq_case.py $C 12 > gpurun_out/${T}_stats_${C}.log 2>&1 || { echo "stats $C failed"; tail -5 gpurun_out/${T}_stats_${C}.log; exit 1; }
  echo "stats $C ok"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${T}_sq1_${C} -- python3 tools/sq_case.py $C 3 > gpurun_out/${T}_sq1_${C}.log 2>&1 || { echo "sq1 $C failed"; tail -5 gpurun_out/${T}_sq1_${C}.log; exit 1; }
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/${T}_sq2_${C} -- python3 tools/sq_case.py $C 3 > gpurun_out/${T}_sq2_${C}.log 2>&1 || { echo "sq2 $C failed"; tail -5 gpurun_out/${T}_sq2_${C}.log; exit 1; }
  echo "sq $C ok"
done
python3 tools/r04_collect.py $T $CASES
# the overlap question: where K0-ahead actually runs (timestamps), and the plain step beside it
for C in ahead c1; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace_${C} -- python3 tools/sq_case.py $C 40 > gpurun_out/${T}_trace_${C}.log 2>&1 && python3 tools/overlap_trace.py gpurun_out/${T}_trace_${C} 4 > gpurun_out/${T}_overlap_${C}.txt
done
tail -20 gpurun_out/${T}_overlap_ahead.txt
