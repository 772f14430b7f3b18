cd $GRAFT_REPO_ROOT
bash tools/shape_trace.sh r3m halfcheetah | head -24
python -m pytest tests/test_hip_parity.py -q -x -k "one_block_kernels and 31" 2>&1 | tail -2
for t in "31=0" "31=1,32=2,33=8" "31=1,32=2,33=16" "31=1,32=3,33=8" "31=1,32=2,33=0" "31=1,32=0,33=12"; do
  bash tools/ab_trace.sh r3m "$t" "$t" 2>&1 | grep -E "^== |gemm_tn_|one update step" | head -3
done
