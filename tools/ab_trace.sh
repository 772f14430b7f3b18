# kernel timeline of one serial update step (bench.py's roofline pass) for two knob settings
# usage: tools/ab_trace.sh <tag> "<tune A>" "<tune B>"   e.g. tools/ab_trace.sh r3e "31=1" "31=0"
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-ab}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for t in "$2" "$3"; do
  i=$((i+1))
  args=""; for kv in $(echo $t | tr ',' ' '); do args="$args --tune $kv"; done
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$i -- python3 $R/bench.py --steps 6 --warmup 2 --passes 1 --spin-up 50 --no-cpu-baseline --no-fp32 --no-pixel --no-configs $args > $O/trace$i.log 2>&1
  python3 $R/tools/trace_step.py $(ls $O/trace$i/*/*kernel_trace.csv) v > $O/step_serial_$i.txt
  python3 $R/tools/trace_step.py $(ls $O/trace$i/*/*kernel_trace.csv) v 5 > $O/step_overlapped_$i.txt
  rm -rf $O/trace$i
  echo "== $t (serial pass)"; head -24 $O/step_serial_$i.txt
done
