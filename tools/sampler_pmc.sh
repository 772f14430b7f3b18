# GPU box: HBM-side traffic of the split sampler per launch (rocprofv3 PMC passes, one counter per run as the guide prescribes),
# reduced by tools/pmc_traffic.py into profiles/pmc_traffic_probe5_bf16.json, which `bench.py --probe 5` reports as `traffic`.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sampler_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --spin-up 0 --n-steps 60 --no-cpu-baseline --no-pixel --no-fp32 --probe 5 > $O/pmc_fetch.log 2>&1 && echo fetch ok
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 1 --spin-up 0 --n-steps 60 --no-cpu-baseline --no-pixel --no-fp32 --probe 5 > $O/pmc_write.log 2>&1 && echo write ok
cd $R
python3 tools/pmc_traffic.py $(ls $O/pmc_fetch/*/*counter_collection.csv) $(ls $O/pmc_write/*/*counter_collection.csv) sample_chain_split_kernel $O/pmc_traffic_probe5_bf16.json > $O/pmc_traffic.txt
cat $O/pmc_traffic.txt | tail -5; cat $O/pmc_traffic_probe5_bf16.json
rm -rf $O/pmc_fetch $O/pmc_write
