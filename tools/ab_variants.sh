# GPU box: A/B of library variants (dppo_amd/lib/libdppo_hip_<name>.so; "" = the default build).
# usage: tools/ab_variants.sh <tag> <name> [<name> ...]     ("default" = libdppo_hip.so)
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset DPPO_HIP_LIB; else export DPPO_HIP_LIB=$R/dppo_amd/lib/libdppo_hip_$v.so; fi
  if [ $rep = 1 ]; then timeout -k 10 200 python3 tools/fused_bench.py > $O/fused_$v.txt 2>&1; grep -v amdgpu.ids $O/fused_$v.txt | tail -n 8; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pixel --no-fp32 > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-14s value %.2f M samples/s  step %.4f ms  sampler %.4f ms" % (sys.argv[2], d['value']/1e6, d['ms_per_step'], d['sampler_ms_per_call']), flush=True)
PY
done; done
