"""Copies the latest gpurun_out rocprof summaries into profiles/ (tracked) and rebuilds
profiles/pmc_traffic.json (HBM bytes and SQ instruction counts per launch of the headline kernels).
usage: python tools/refresh_profiles.py TAG      (after `bash tools/prof_round.sh TAG` on a gpurun box)"""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
def one(d, pat): return max(glob.glob(os.path.join(R, 'gpurun_out', d, '**', pat), recursive=True), key=os.path.getmtime)
def avg(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter: agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}
shutil.copy(one(f'{tag}_stats', '*kernel_stats.csv'), f'{R}/profiles/{tag}_kernel_stats.csv')          # the headline alone
try: shutil.copy(one(f'{tag}_stats_other', '*kernel_stats.csv'), f'{R}/profiles/{tag}_other_kernel_stats.csv')   # other_configs, small batches, host path
except Exception: pass
for suffix in ('bench.json', 'bench_g2.json', 'bench_g2_strong.json'):
    src = os.path.join(R, 'gpurun_out', f'{tag}_{suffix}')
    if os.path.exists(src) and os.path.getsize(src): shutil.copy(src, f'{R}/profiles/{tag}_{suffix}')
f = avg(one(f'{tag}_fetch', '*counter_collection.csv'), 'FETCH_SIZE')
w = avg(one(f'{tag}_write', '*counter_collection.csv'), 'WRITE_SIZE')
sqf = one(f'{tag}_sq', '*counter_collection.csv')
sq = {c: avg(sqf, c) for c in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY')}
names = {'k_prepare': 'k_prepare_stereo', 'k_autocorr': 'k_autocorr_wt', 'k_encode': 'k_encode_pow2'}
old = json.load(open(f'{R}/profiles/pmc_traffic.json'))
sys.path.insert(0, R)
from flake_amd.srcid import kernel_sources_sha1
out = {"_note": old["_note"], "_calibration": old["_calibration"], "_tag": tag,
       "_sq_note": "valu_per_wave = SQ_INSTS_VALU / SQ_WAVES of the same batch (one SQ pass, rocprofv3 --pmc with --kernel-trace only); wave_cycles = 4 x SQ_WAVE_CYCLES / SQ_WAVES",
       "_src_sha1": kernel_sources_sha1()}     # the kernel sources these counters were measured with
def pick(d, sym): return next((v for k, v in d.items() if sym in k), 0.0)
for short, sym in names.items():
    fk, wk = pick(f, sym), pick(w, sym)
    waves = pick(sq['SQ_WAVES'], sym)
    ent = {"symbol": sym, "workload": "configs[1]", "frames": 4096, "FETCH_SIZE_KiB_raw": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
           "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    if waves:
        ent.update({"waves_per_launch": int(waves), "valu_per_wave": round(pick(sq['SQ_INSTS_VALU'], sym) / waves, 1),
                    "salu_per_wave": round(pick(sq['SQ_INSTS_SALU'], sym) / waves, 1),
                    "lds_per_wave": round(pick(sq['SQ_INSTS_LDS'], sym) / waves, 1),
                    "wave_cycles": round(4 * pick(sq['SQ_WAVE_CYCLES'], sym) / waves, 0),
                    "wait_any_cycles": round(4 * pick(sq['SQ_WAIT_ANY'], sym) / waves, 0)})
    out[short] = ent
json.dump(out, open(f'{R}/profiles/pmc_traffic.json', 'w'), indent=1)
for d, n in ((f'{tag}_fetch', f'{tag}_pmc_fetch_size.csv'), (f'{tag}_write', f'{tag}_pmc_write_size.csv'), (f'{tag}_sq', f'{tag}_pmc_sq.csv')):
    rows = list(csv.DictReader(open(one(d, '*counter_collection.csv'))))
    with open(f'{R}/profiles/{n}', 'w') as fo:
        fo.write("Kernel_Name,Counter_Name,Counter_Value,Grid_Size,LDS_Block_Size,VGPR_Count\n")
        for r in rows:
            if 'fhip' in r['Kernel_Name']:
                fo.write(f"\"{r['Kernel_Name']}\",{r['Counter_Name']},{r['Counter_Value']},{r['Grid_Size']},{r['LDS_Block_Size']},{r['VGPR_Count']}\n")
print(json.dumps({k: v for k, v in out.items() if not k.startswith('_')}, indent=1))
