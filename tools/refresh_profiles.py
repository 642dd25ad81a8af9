"""Copies the latest gpurun_out rocprof summaries into profiles/ (tracked).
usage: python tools/refresh_profiles.py TAG STATS_DIR FETCH_DIR WRITE_DIR BENCH_JSON"""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats, fetch, write, bench = sys.argv[1:6]
def one(d, pat): return max(glob.glob(os.path.join(R, 'gpurun_out', d, '*', pat)), key=os.path.getmtime)   # the newest run
def avg(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter: agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}
shutil.copy(one(stats, '*kernel_stats.csv'), f'{R}/profiles/{tag}_kernel_stats.csv')
shutil.copy(os.path.join(R, 'gpurun_out', bench), f'{R}/profiles/{tag}_bench.json')
f = avg(one(fetch, '*counter_collection.csv'), 'FETCH_SIZE')
w = avg(one(write, '*counter_collection.csv'), 'WRITE_SIZE')
names = {'k_prepare': 'k_prepare_stereo', 'k_autocorr': 'k_autocorr_wt', 'k_encode': 'k_encode_pow2'}
old = json.load(open(f'{R}/profiles/pmc_traffic.json'))
sys.path.insert(0, R)
from flake_amd.srcid import kernel_sources_sha1
out = {"_note": old["_note"], "_calibration": old["_calibration"], "_tag": tag,
       "_src_sha1": kernel_sources_sha1()}     # the kernel sources these counters were measured with
for short, sym in names.items():
    fk = next((v for k, v in f.items() if sym in k), 0.0)
    wk = next((v for k, v in w.items() if sym in k), 0.0)
    out[short] = {"symbol": sym, "FETCH_SIZE_KiB_raw": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                  "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
json.dump(out, open(f'{R}/profiles/pmc_traffic.json', 'w'), indent=1)
for d, n in ((fetch, f'{tag}_pmc_fetch_size.csv'), (write, f'{tag}_pmc_write_size.csv')):
    rows = list(csv.DictReader(open(one(d, '*counter_collection.csv'))))
    with open(f'{R}/profiles/{n}', 'w') as fo:
        fo.write("Kernel_Name,Counter_Name,Counter_Value,Grid_Size,LDS_Block_Size,VGPR_Count\n")
        for r in rows:
            if 'fhip' in r['Kernel_Name']:
                fo.write(f"\"{r['Kernel_Name']}\",{r['Counter_Name']},{r['Counter_Value']},{r['Grid_Size']},{r['LDS_Block_Size']},{r['VGPR_Count']}\n")
print(json.dumps({k: v for k, v in out.items() if not k.startswith('_')}, indent=1))
