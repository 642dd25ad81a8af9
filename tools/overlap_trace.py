"""Reads a rocprofv3 --kernel-trace CSV of `tools/sq_case.py ahead` (or c1) and prints where each kernel of the
last steps ran: start / end relative to the step's first kernel, and for K0 how much of its span lay inside
K1's and K3's windows.   python tools/overlap_trace.py DIR [steps_to_show]"""
import csv, glob, re, sys
d = sys.argv[1]
show = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if not m:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
def short(k):
    return "K0" if "prepare" in k else "K1" if "autocorr" in k else "K2" if "lpc" in k else "K3" if "encode" in k else k
enc = [r for r in rows if short(r[2]) == "K3"]
t_first = enc[-show - 1][1] if len(enc) > show else rows[0][0]
sel = [r for r in rows if r[0] >= t_first]
t0 = sel[0][0]
for s, e, k, q, st in sel:
    print(f"{short(k):3s} q{q:>3s} s{st:>3s}  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us   ({(e - s) / 1e3:7.1f} us)")
def inter(a, b):
    return max(0, min(a[1], b[1]) - max(a[0], b[0]))
k0 = [r for r in sel if short(r[2]) == "K0"]
for kk in ("K1", "K3"):
    tot = sum(inter(a, b) for a in k0 for b in sel if short(b[2]) == kk)
    print(f"K0 time inside {kk} windows: {tot / 1e3 / max(1, len(k0)):.1f} us per K0 launch (K0 avg span {sum(e - s for s, e, *_ in k0) / 1e3 / max(1, len(k0)):.1f} us)")
step = (enc[-1][1] - enc[-show - 1][1]) / show / 1e3 if len(enc) > show else 0
print(f"step (K3 end to K3 end): {step:.1f} us")
