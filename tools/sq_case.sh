# SQ counters of one named workload's kernels (two --pmc passes, --kernel-trace only).  bash tools/sq_case.sh CASE
C=${1:-l8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/sqc_${C}_1 -- python3 tools/sq_case.py $C > gpurun_out/sqc_${C}_1.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/sqc_${C}_2 -- python3 tools/sq_case.py $C > gpurun_out/sqc_${C}_2.log 2>&1 &&
python - $C <<'PY'
import csv,glob,collections,re,sys
for d in (1,2):
    f=glob.glob("gpurun_out/sqc_%s_%d/**/*_counter_collection.csv"%(sys.argv[1],d),recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
    for r in csv.DictReader(open(f)):
        m=re.search(r"(k_\w+(<[^>]*>)?)",r["Kernel_Name"])
        if not m: continue
        agg[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[m.group(1)]+=1
    for k,g in agg.items():
        w=g["SQ_WAVES"] or 1
        print(sys.argv[1], k, "x%d"%n[k], "waves/launch %d"%(w/n[k]), {c: round(v/w,1) for c,v in g.items() if c!="SQ_WAVES"})
PY
