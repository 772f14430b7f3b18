set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
DPPO_HIP_LIB=$R/dppo_amd/lib/libdppo_hip_stamps.so timeout -k 10 200 python3 tools/fused_bench.py --stamps > $O/stamps.txt 2>&1 || tail -5 $O/stamps.txt
sed -i 's/--no-cpu-baseline \$t/--no-cpu-baseline --no-pixel --no-fp32 $t/' tools/ab_bench.sh
bash tools/ab_bench.sh "" "--tune 5=1" "--tune 5=2" "--tune 5=4" "--tune 5=-1" "--graph" "--tune 3=512" "--tune 3=384"
