cd $GRAFT_REPO_ROOT
for t in "31=1" "31=1,34=1" "31=1,34=2" "31=1,34=4" "31=1,34=6" "31=1,34=3" "31=1,34=7"; do
  bash tools/ab_trace.sh r3f "$t" "$t" 2>&1 | grep -E "^== |gemm_tn_fragl" | head -2
done
