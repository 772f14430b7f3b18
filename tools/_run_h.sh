set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pixel --no-fp32 --tune 10=2 > $O/trace.log 2>&1
cd $R
python3 tools/trace_step.py $(ls $O/trace/*/*kernel_trace.csv) v 5 > $O/step_overlapped_actor_first.txt
rm -rf $O/trace
cat $O/step_overlapped_actor_first.txt
