# hardware counters of one kernel over bench.py's serial pass: tools/pmc_kernel.sh <tag> <kernel substring> "<tune>" 
# (separate --pmc passes, no trace flags beside them; summaries land in gpurun_out/<tag>/)
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; K=$2; O=$R/gpurun_out/$TAG; mkdir -p $O
args=""; for kv in $(echo $3 | tr ',' ' '); do args="$args --tune $kv"; done
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc$i -- python3 $R/bench.py --steps 4 --warmup 2 --passes 1 --spin-up 20 --no-cpu-baseline --no-fp32 --no-pixel --no-configs --tune 2=0 $args > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/pmc$i.log; continue; }
  python3 - "$(ls $O/pmc$i/*/*counter_collection.csv)" "$K" <<'PY'
import csv,sys,collections
vals=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        vals[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
for c,d in vals.items():
    v=sorted(d.values()); n=len(v)
    print(f"{c:40s} n={n:3d} min {v[0]:.4g} median {v[n//2]:.4g} max {v[-1]:.4g}")
PY
  rm -rf $O/pmc$i
done
