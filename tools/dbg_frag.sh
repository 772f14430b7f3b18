cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_full_size.py -q -x 2>&1 | tail -2
bash tools/ab_trace.sh r3p "" "" 2>&1 | grep -v "^ *[0-9]" | head -16
head -40 gpurun_out/r3p/step_overlapped_1.txt | tail -22
