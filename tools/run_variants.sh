for v in "" var_nw "" var_nw; do
  if [ -n "$v" ]; then export FHIP_LIB=$PWD/flake_amd/lib/$v.so; else unset FHIP_LIB; fi
  python tools/c34_probe.py 2>&1 | grep -E "c4|lvl5" | sed "s/^/[$v] /"
  timeout -k 10 200 python tools/srch_probe.py 2>&1 | grep -E "level8" | sed "s/^/[$v] /"
done
