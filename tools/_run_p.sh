set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; echo TESTS_FAILED; }
tail -n 3 $O/tests.log
timeout -k 10 200 python3 tools/fused_bench.py > $O/fused.txt 2>&1; grep -v amdgpu.ids $O/fused.txt | tail -n 5
bash tools/ab_bench.sh "" ""
