# GPU box: the reduction launch behind the weight-gradient GEMMs (tail_reduce_kernel) with parts of it switched off (knob 8 bits 2-4):
# 0 = all, 24 = slab reductions only, 20 = bias sums only, 12 = loss statistics only.  rocprofv3 average per launch.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s3x; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for v in 0 24 20 12; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$v -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pixel --no-fp32 --tune 2=0 --tune 8=$v > $O/log_$v.txt 2>&1
  echo "knob8=$v"; grep -h "tail_reduce\|post_reduce" $O/st_$v/*/*kernel_stats.csv | cut -c1-120
done
