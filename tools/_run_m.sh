set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; echo TESTS_FAILED; }
tail -n 3 $O/tests.log
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pixel --no-fp32 > $O/bench.json 2> $O/bench.err; python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.2f M samples/s  step %.4f ms  sampler %.4f ms  env-steps %.2f M" % (d['value']/1e6, d['ms_per_step'], d['sampler_ms_per_call'], d['env_steps_per_sec']/1e6))
PY
timeout -k 10 300 python3 tools/shape_bench.py > $O/shapes.txt 2>&1; grep -v amdgpu.ids $O/shapes.txt | tail -n 6
