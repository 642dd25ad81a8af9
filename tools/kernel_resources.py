"""Register / scratch / occupancy table of every kernel, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage; runs without a GPU).
python tools/kernel_resources.py [file.hip ...] > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from flake_amd import build as fb
files = sys.argv[1:] or [s for s in fb.HIP_SRCS if "api" not in s]
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s}")
for f in files:
    src = os.path.join(fb.PKG, f) if not os.path.isabs(f) else f
    cmd = [fb.HIPCC, *[x for x in fb.HIP_FLAGS if x != "-shared"], "-I", os.path.join(R, "include"),
           "-I", os.path.join(fb.PKG, "csrc"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    def flush():
        if cur.get("name"):
            nm = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            nm = re.sub(r"\(anonymous namespace\)::|fhip::|void ", "", nm)
            nm = re.sub(r"\(.*", "", nm)
            print(f"{nm[:70]:70s} {cur.get('VGPRs','?'):>5s} {cur.get('AGPRs','?'):>5s} {cur.get('SGPRs','?'):>5s} "
                  f"{cur.get('VGPRs Spill','?'):>6s} {cur.get('SGPRs Spill','?'):>6s} {cur.get('ScratchSize [bytes/lane]','?'):>7s} "
                  f"{cur.get('Occupancy [waves/SIMD]','?'):>4s} {cur.get('LDS Size [bytes/block]','?'):>6s}")
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?): +(\S+) \[-Rpass", line) or re.search(r"remark: +(.*?): +(\S+) \[-Rpass", line)
        if not m:
            m2 = re.search(r":\d+:\d+: remark: (Function Name|Name): (\S+)", line)
            if m2:
                flush(); cur = {"name": m2.group(2)}
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k in ("Function Name", "Name"):
            flush(); cur = {"name": v}
        else:
            cur[k] = v
    flush()
