for lib in libflakehip.so libflakehip_noprod.so libflakehip_nowalk.so; do for sp in 0 1; do
# the probe libraries: `python -m flake_amd.build probes` (here, before gpurun: built .so files travel); a missing one is skipped
if [ $sp = 0 ]; then export FHIP_NO_LAG_SPLIT=1; else unset FHIP_NO_LAG_SPLIT; fi
FHIP_LIB=$PWD/flake_amd/lib/$lib python bench.py --frames 512 --no-cpu-baseline --no-other-configs --steps 200 --warmup 50 > gpurun_out/r03_k1p.json 2> gpurun_out/r03_k1p.err
python -c "
import json;d=json.load(open('gpurun_out/r03_k1p.json'));print('$lib split=$sp',d['ms_per_step'],d['roofline']['kernel_ms'])" || tail -3 gpurun_out/r03_k1p.err
done; done
