"""Summarise an SQ-counter pass over tools/ablate.py: per ablation variant, per-wave
instruction counts and wave-cycle split of the encode kernel.
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU \
             SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d DIR -- python3 tools/ablate.py
   python tools/pmc_k3.py DIR [kernel-prefix]"""
import csv, collections, re, glob, sys
d = sys.argv[1]; pref = sys.argv[2] if len(sys.argv) > 2 else "k_encode"
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
disp = {}
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if not m: continue
    disp.setdefault(int(r["Dispatch_Id"]), {"k": m.group(1)})[r["Counter_Name"]] = float(r["Counter_Value"])
enc = [v for _, v in sorted(disp.items()) if v["k"].startswith(pref)]
for i in range(0, len(enc), 12):
    g = enc[min(i + 11, len(enc) - 1)]; w = g["SQ_WAVES"]
    print("%2d %-18s waves %6d  VALU/wave %5.0f  SALU %5.0f  LDS %4.0f  wave-cycles %6.0f  wait_any %6.0f  wait_inst %6.0f" % (
        i // 12, g["k"], w, g["SQ_INSTS_VALU"] / w, g["SQ_INSTS_SALU"] / w, g["SQ_INSTS_LDS"] / w,
        4 * g["SQ_WAVE_CYCLES"] / w, 4 * g["SQ_WAIT_ANY"] / w, 4 * g["SQ_WAIT_INST_ANY"] / w))
