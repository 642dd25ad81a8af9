"""Folds tools/r04_measure.sh's passes into profiles/: per config the rocprofv3 kernel stats
(profiles/<TAG>_<case>_kernel_stats.csv) and one JSON of SQ counters per kernel instance and wave
(profiles/<TAG>_sq_counters.json).  python tools/r04_collect.py TAG case [case ...]"""
import collections, csv, glob, json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cases = sys.argv[1], sys.argv[2:]
out = {}
for c in cases:
    try:
        f = max(glob.glob(f"{R}/gpurun_out/{tag}_stats_{c}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
        shutil.copy(f, f"{R}/profiles/{tag}_{c}_kernel_stats.csv")
    except Exception as e:
        print("no stats for", c, e)
    ent = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(set)
    for q in (1, 2):
        fs = glob.glob(f"{R}/gpurun_out/{tag}_sq{q}_{c}/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            name = r["Counter_Name"] + ("" if q == 1 or r["Counter_Name"] != "SQ_WAVES" else "_pass2")
            ent[k][name] += float(r["Counter_Value"])
            nd[(k, q)].add(r["Dispatch_Id"])
            ent[k]["_vgpr"] = float(r.get("VGPR_Count") or 0)
            ent[k]["_lds"] = float(r.get("LDS_Block_Size") or 0)
            ent[k]["_wg"] = float(r.get("Workgroup_Size") or 0)
    rows = {}
    for k, g in ent.items():
        w1 = g.get("SQ_WAVES") or 0
        w2 = g.get("SQ_WAVES_pass2") or 0
        row = {"launches": len(nd[(k, 1)]), "waves_per_launch": round(w1 / max(1, len(nd[(k, 1)])), 1),
               "vgprs": g["_vgpr"], "lds_bytes": g["_lds"], "workgroup": g["_wg"]}
        for cn, v in g.items():
            if cn.startswith("_") or cn.startswith("SQ_WAVES"):
                continue
            w = w2 if cn in ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY",
                             "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_SMEM") else w1
            row[cn + "_per_wave"] = round(v / w, 1) if w else None
        rows[k] = row
    out[c] = rows
# profiles/pmc_cases.json: what bench.py's other_configs rows price their kernels with (per_kernel bounds)
sys.path.insert(0, R)
from flake_amd.srcid import kernel_sources_sha1
STEPS = 3          # tools/r04_measure.sh runs the SQ passes over three steps of the case
pc = {"_note": "per workload case (tools/sq_case.py) and kernel instance: SQ counters per wave of one rocprofv3 --pmc pass "
               "(--kernel-trace only beside it), cycles = 4 x the quad-cycle counters; launches_per_step = dispatches / steps",
      "_tag": tag, "_src_sha1": kernel_sources_sha1()}
try:
    old = json.load(open(f"{R}/profiles/pmc_cases.json"))
    if old.get("_src_sha1") == pc["_src_sha1"]:
        pc.update({k: v for k, v in old.items() if not k.startswith("_")})      # cases measured in an earlier call
except Exception:
    pass
for c, rows in out.items():
    pc[c] = {}
    for k, r in rows.items():
        if not r.get("SQ_INSTS_VALU_per_wave"):
            continue
        pc[c][k] = {"launches_per_step": round(r["launches"] / STEPS, 2), "waves_per_launch": r["waves_per_launch"],
                    "valu_per_wave": r["SQ_INSTS_VALU_per_wave"], "salu_per_wave": r.get("SQ_INSTS_SALU_per_wave"),
                    "lds_per_wave": r.get("SQ_INSTS_LDS_per_wave"), "mfma_per_wave": r.get("SQ_INSTS_MFMA_per_wave") or 0.0,
                    "mfma_busy_cycles_per_wave": r.get("SQ_VALU_MFMA_BUSY_CYCLES_per_wave") or 0.0,
                    "wave_cycles": round(4 * (r.get("SQ_WAVE_CYCLES_per_wave") or 0)),
                    "wait_any_cycles": round(4 * (r.get("SQ_WAIT_ANY_per_wave") or 0)),
                    "wait_inst_cycles": round(4 * (r.get("SQ_WAIT_INST_ANY_per_wave") or 0)),
                    "vgprs": r.get("vgprs")}
json.dump(pc, open(f"{R}/profiles/pmc_cases.json", "w"), indent=1)
p = f"{R}/profiles/{tag}_sq_counters.json"
json.dump({"_note": "per kernel instance: counter totals / SQ_WAVES of the same pass (quad-cycle counters as read: x4 = cycles); "
                    "tools/r04_measure.sh, sq_case.py cases", **out}, open(p, "w"), indent=1)
for c, rows in out.items():
    for k, r in rows.items():
        print(c, k, json.dumps(r))
