"""Measurement probe: the kernels of the LAST step of a rocprofv3 --kernel-trace run, in start order, with the queue each ran
on and the gaps between -- where a ragged batch's wall time goes.   python tools/step_timeline.py <trace dir> <first kernel name>"""
import csv, glob, sys
d, first = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
end = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("fhip::(anonymous namespace)::", "").replace("void ", "")[:48]
    print(f"{s/1000:8.1f} {e/1000:8.1f} {(e-s)/1000:7.1f} q{r['Queue_Id']:>3} grid {r['Grid_Size_X']:>9} {name}")
    end = max(end, e)
print("step", end / 1000, "us")
