cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
python -m pytest tests/test_hip_parity.py tests/test_bf16_parity.py tests/test_full_size.py -q -x > gpurun_out/r3i/t1.log 2>&1; tail -4 gpurun_out/r3i/t1.log
for t in "35=1" "35=0"; do
  echo "== $t"; python tools/shape_bench.py --shapes hopper,halfcheetah,can --tune $t 2>&1 | grep -v amdgpu.ids
done
bash tools/ab_trace.sh r3i "35=1" "35=0" | grep -v "^ *[0-9]" | head -40
