"""Measurement probe (not part of the product): bench.py's configs[4] rows alone -- ms per batch and the serial per-kernel sums.
python tools/vbs_time.py [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for level, nblk in ((10, 1024), (12, 1024), (10, 8192), (12, 8192)):
    r = bench.vbs_case(0, level, nblk, steps if nblk == 1024 else max(steps // 3, 5), cpu=False)
    print(json.dumps({"level": level, "blocks": nblk, "ms": r["ms_per_step"], "frames": r["frames_out"], "bytes": r["bytes_out"],
                      "kernels": r["kernel_ms_serial_sums"]}), flush=True)
