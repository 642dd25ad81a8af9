# SQ LDS counters of K1 per probe build (who causes the bank conflicts): rocprofv3 --pmc with --kernel-trace only
# the probe libraries: `python -m flake_amd.build probes` (here, before gpurun: built .so files travel); a missing one is skipped
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in libflakehip.so libflakehip_noprod.so libflakehip_nowalk.so libflakehip_nob.so; do
export FHIP_LIB=$PWD/flake_amd/lib/$lib
[ -f $FHIP_LIB ] || { echo "$lib missing: python -m flake_amd.build probes"; continue; }
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/r03_sq_$lib -- python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --profile-steps 0 --no-cpu-baseline --no-other-configs > gpurun_out/r03_sq_$lib.log 2>&1
python - <<PY
import csv,glob,collections,re
f=glob.glob("gpurun_out/r03_sq_$lib/**/*_counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_\w+)",r["Kernel_Name"])
    if not m: continue
    agg[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); n[m.group(1)]+=1
for k,g in agg.items():
    if 'autocorr' not in k: continue
    w=g["SQ_WAVES"] or 1
    print("$lib", k, "x%d"%n[k], {c: round(v/w,1) for c,v in g.items() if c!="SQ_WAVES"})
PY
done
