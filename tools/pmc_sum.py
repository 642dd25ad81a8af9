"""Summarise an SQ-counter pass: per kernel name, per-wave instruction counts and the
wave-cycle split, averaged over its dispatches.
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU \
             SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d DIR -- python3 <script>
   python tools/pmc_sum.py DIR"""
import csv, collections, re, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
disp = {}
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if not m: continue
    disp.setdefault(int(r["Dispatch_Id"]), {"k": m.group(1)})[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for _, v in disp.items():
    if "SQ_WAVES" not in v: continue
    cnt[v["k"]] += 1
    for k, x in v.items():
        if k != "k": agg[v["k"]][k] += x
for k in sorted(agg):
    g = agg[k]; w = g["SQ_WAVES"]
    print("%-22s x%-4d waves %7d  VALU/wave %6.0f  SALU %5.0f  LDS %5.0f  wave-cycles %7.0f  wait_any %6.0f  wait_inst %6.0f" % (
        k, cnt[k], w / cnt[k], g["SQ_INSTS_VALU"] / w, g["SQ_INSTS_SALU"] / w, g["SQ_INSTS_LDS"] / w,
        4 * g["SQ_WAVE_CYCLES"] / w, 4 * g["SQ_WAIT_ANY"] / w, 4 * g["SQ_WAIT_INST_ANY"] / w))
