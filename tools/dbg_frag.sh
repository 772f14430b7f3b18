cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_bf16_parity.py -q -x 2>&1 | tail -2
for s in square transport furniture; do bash tools/shape_trace.sh r3o $s | head -14; done
python tools/shape_bench.py 2>&1 | grep -v amdgpu
