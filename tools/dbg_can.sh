cd $GRAFT_REPO_ROOT
for t in "" "11=0" "22=3" "25=0" "23=0"; do
  echo "== DPPO_TUNE=$t"
  DPPO_TUNE=$t python -m pytest "tests/test_hip_parity.py::test_recomputed_logprobs_equal_precomputed_ones_for_every_kernel_family" -q -k "bf16" 2>&1 | grep -E "passed|failed|Obtained|FAILED" | head -8
done
