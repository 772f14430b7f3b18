# GPU box: the artefacts of round 3's final build (copied from gpurun_out/r03fin1 into profiles/ as r03_final_* afterwards).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT_TAG:-r03fin5}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[1] bench full"; timeout -k 10 500 python3 $R/bench.py > $O/bench.json 2> $O/bench.err && tail -c 400 $O/bench.json && echo
echo "[2] rocprof stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pixel --no-fp32 > $O/stats.log 2>&1 && echo ok
echo "[2b] rocprof stats, side streams off"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pixel --no-fp32 --tune 2=0 > $O/stats_serial.log 2>&1 && echo ok
echo "[3] pmc fetch"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --n-steps 60 --no-cpu-baseline --no-pixel --no-fp32 > $O/pmc_fetch.log 2>&1 && echo ok
echo "[4] pmc write"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 1 --n-steps 60 --no-cpu-baseline --no-pixel --no-fp32 > $O/pmc_write.log 2>&1 && echo ok
echo "[4b] mfma util"; timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 4 --warmup 1 --n-steps 60 --no-cpu-baseline --no-pixel --no-fp32 > $O/pmc_mfma.log 2>&1 && echo ok
echo "[4c] timeline"; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --passes 1 --spin-up 50 --no-cpu-baseline --no-pixel --no-fp32 --no-configs > $O/trace.log 2>&1 && echo ok
cd $R
python3 tools/pmc_traffic.py $(ls $O/pmc_fetch/*/*counter_collection.csv) $(ls $O/pmc_write/*/*counter_collection.csv) gemm_tn_group_kernel $O/pmc_traffic_probe2g_bf16.json > $O/pmc_traffic.txt
python3 tools/mfma_util.py $(ls $O/pmc_mfma/*/*counter_collection.csv) $O/mfma_util.json > $O/mfma_util.txt
python3 tools/trace_step.py $(ls $O/trace/*/*kernel_trace.csv) v 5 > $O/step_overlapped.txt
python3 tools/trace_step.py $(ls $O/trace/*/*kernel_trace.csv) v > $O/step_serial.txt || true
cp $(ls $O/stats/*/*kernel_stats.csv) $O/bench_kernel_stats.csv
cp $(ls $O/stats_serial/*/*kernel_stats.csv) $O/bench_serial_kernel_stats.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/trace $O/stats $O/stats_serial
echo "[5] fused bench"; timeout -k 10 200 python3 tools/fused_bench.py > $O/fused_bench.txt 2>&1 && tail -6 $O/fused_bench.txt
echo "[6] shapes"; timeout -k 10 300 python3 tools/shape_bench.py > $O/shapes.txt 2>&1 && tail -5 $O/shapes.txt
echo "[7] dp2 rehearsal through the launcher"; HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 --passes 3 --spin-up 20 --backend gloo --share-gpu --no-cpu-baseline --no-pixel --no-fp32 --n-steps 50 > $O/dp2.json 2> $O/dp2.err && tail -c 300 $O/dp2.json
echo "[8] sampler kernel probe"; timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pixel --no-fp32 --probe 5 > $O/bench_probe5.json 2> $O/bench_probe5.err && tail -c 600 $O/bench_probe5.json
echo "[9] sampler, one workgroup per tile (knob 27 = 0)"; timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pixel --no-fp32 --probe 5 --tune 27=0 > $O/bench_probe5_nosplit.json 2> $O/bench_probe5_nosplit.err && tail -c 600 $O/bench_probe5_nosplit.json
if [ -f dppo_amd/lib/libdppo_hip_stamps.so ]; then echo "[10] split sampler stamps"; DPPO_HIP_LIB=$R/dppo_amd/lib/libdppo_hip_stamps.so timeout -k 10 120 python3 tools/sampler_stamps.py > $O/split_sampler_stamps.txt 2>/dev/null; cat $O/split_sampler_stamps.txt; fi

echo "[11] kernel registers"; python3 tools/kernel_regs.py dppo_amd/csrc/fused.hip forward_merged backward_one > $O/one_block_kernel_regs.txt 2>&1; tail -3 $O/one_block_kernel_regs.txt
echo "[12] vision kernel stats"; cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vstats -- python3 $R/tools/vision_bench.py --once --kind unet > $O/vstats.log 2>&1 && cp $(ls $O/vstats/*/*kernel_stats.csv) $O/vision_kernel_stats.csv; rm -rf $O/vstats; cd $R
