"""Static vector-instruction mix of every kernel -> profiles/valu_mix.json: how many of a kernel's VALU instructions
are of the classes tools/ubench_valu.hip measured at HALF the SIMD-32 issue rate on gfx950 (4 cycles per wave64
instruction with several waves per SIMD: three-operand integer forms, v_lshlrev, min/max, compares, cndmask, carries,
multiplies, dot products, alignbyte / alignbit / perm / bfe, DPP forms, read/writelane, every fp64 instruction) against
the full-rate ones (2 cycles: v_add_u32, v_sub_u32, v_ashrrev_i32, v_and / v_or / v_xor, v_mov_b32, fp32).  Opcodes
nobody measured count as full rate (a bound must not be priced too high).  A STATIC count (no trip counts): an estimate
of the dynamic mix for these mostly unrolled kernels.  bench.py prices `valu_issue` bounds with avg_cycles.
    python tools/valu_mix.py [DIR with *.s | none: compiles flake_amd/csrc/*.hip with -S into /tmp/isa]"""
import glob, json, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from flake_amd import build as fb
from flake_amd.srcid import kernel_sources_sha1
d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa"
os.makedirs(d, exist_ok=True)
jobs = []
for s in fb.HIP_SRCS:
    if "api" in s:
        continue
    src = os.path.join(fb.PKG, s)
    out = os.path.join(d, os.path.basename(s).replace(".hip", ".s"))
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), *[os.path.getmtime(os.path.join(fb.PKG, h)) for h in fb.HIP_HDRS]):
        cmd = [fb.HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(R, "include"),
               "-I", os.path.join(fb.PKG, "csrc"), "-S", "--cuda-device-only", "-o", out, src]
        jobs.append(subprocess.Popen(cmd, stderr=subprocess.DEVNULL))
for j in jobs:
    j.wait()
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32",
        "v_add_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_sub_f32", "v_lshrrev_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32",
        "v_pk_mov_b32", "v_mov_b64", "v_nop"}
HALF_RE = re.compile(r"^v_(lshlrev_b32|lshl_add_u32|add3_u32|xad_u32|lshl_or_b32|and_or_b32|or3_b32|bfe_|bfi_|alignb|perm_b32|sad_|mul_|mad_|"
                     r"ffb|bcnt|dot|max_|min_|max3|min3|med3|cmp|cndmask|add_co|addc_co|sub_co|subb_co|subrev_co|readlane|readfirstlane|writelane|"
                     r"cvt_|.*_f64|lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|mbcnt|add_lshl)")
out = {}
for f in sorted(glob.glob(os.path.join(d, "k*.s"))):
    cur = None
    for line in open(f):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            nm = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            nm = re.sub(r"\(anonymous namespace\)::|fhip::|void ", "", nm)
            nm = re.sub(r"\(.*", "", nm)
            cur = out.setdefault(nm, {"valu": 0, "half_rate": 0, "mfma": 0})
            continue
        if line.startswith("\t.amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        if cur is None:
            continue
        t = line.strip().split()
        if not t or not t[0].startswith("v_"):
            continue
        op = t[0]
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            cur["mfma"] += 1
            continue
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        cur["valu"] += 1
        half = ("dpp" in line and "row_" in line) or op.endswith("_dpp") or bool(HALF_RE.match(base))
        if base in FULL and not (op.endswith("_dpp") or "row_" in line):
            half = False
        cur["half_rate"] += 1 if half else 0
for k, v in out.items():
    v["avg_cycles"] = round((2 * (v["valu"] - v["half_rate"]) + 4 * v["half_rate"]) / max(1, v["valu"]), 3)
res = {"_note": __doc__.split("\n    python")[0], "_src_sha1": kernel_sources_sha1(), **out}
json.dump(res, open(os.path.join(R, "profiles", "valu_mix.json"), "w"), indent=1)
for k in sorted(out):
    if any(s in k for s in ("k_encode_pow2<16, 256", "k_order_search<16, 256", "k_prepare_stereo<4, 4, true", "k_autocorr_wt<3, false, 8")):
        print(k, out[k])
