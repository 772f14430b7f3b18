# GPU box: GPU tests, one bench line, and a kernel timeline of one update step (overlapped and serial).
# usage: tools/quick_profile.sh <tag> [pytest args...]
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-quick}; shift || true
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
echo "[1] tests"; timeout -k 10 500 python3 -m pytest tests -m gpu -x -q "$@" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
echo "[2] bench"; timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.2f M samples/s  step %.4f ms  sampler %.4f ms  env-steps %.2f M" % (d['value']/1e6, d['ms_per_step'], d['sampler_ms_per_call'], d['env_steps_per_sec']/1e6))
PY
cd /tmp && export TMPDIR=/tmp
echo "[3] timeline"; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
cd $R
python3 tools/trace_step.py $(ls $O/trace/*/*kernel_trace.csv) v 5 > $O/step_overlapped.txt
python3 tools/trace_step.py $(ls $O/trace/*/*kernel_trace.csv) v > $O/step_serial.txt
rm -rf $O/trace
head -22 $O/step_serial.txt
