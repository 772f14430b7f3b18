set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; echo TESTS_FAILED; }
tail -n 3 $O/tests.log
timeout -k 10 300 python3 tools/shape_bench.py > $O/shapes.txt 2>&1; grep -v amdgpu.ids $O/shapes.txt | tail -n 6
timeout -k 10 300 python3 tools/shape_bench.py --tune 22=0,23=0 > $O/shapes_old.txt 2>&1 || timeout -k 10 300 python3 tools/shape_bench.py --tune 23=0 > $O/shapes_old.txt 2>&1; grep -v amdgpu.ids $O/shapes_old.txt | tail -n 6
