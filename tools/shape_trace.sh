# per-kernel time table of one cfg's update steps: tools/shape_trace.sh <tag> <shape> ["tune"]
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; SH=$2; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$SH -- python3 $R/tools/shape_bench.py --shapes $SH --steps 6 --tune "2=0${3:+,$3}" > $O/tr_$SH.log 2>&1
python3 - "$(ls $O/tr_$SH/*/*kernel_trace.csv)" <<'PY' > $O/kernels_$SH.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
ts=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in rows)
# last 4 update steps: between build_rows kernels
b=[i for i,t in enumerate(ts) if 'build_rows_kernel' in t[2]]
# the update loop's build_rows calls are the last ones before sampler-only phase; take a window of 4 consecutive steps whose segments contain gemm_tn
segs=[]
for x,y in zip(b[:-1],b[1:]):
    seg=ts[x:y]
    if any('gemm_tn' in n for _,_,n in seg) and any('adamw' in n for _,_,n in seg): segs.append(seg)
segs=segs[-4:]
agg=collections.OrderedDict()
for seg in segs:
    for s,e,n in seg:
        n=n.replace('dppo::','').replace('void ','').split('(')[0][:72]
        a=agg.setdefault(n,[0,0.0]); a[0]+=1; a[1]+=(e-s)/1e3
tot=sum(v[1] for v in agg.values())/len(segs)
print(f"{len(segs)} serial update steps, kernel time per step {tot:.1f} us")
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print(f"  {n:72s} x{c/len(segs):4.1f} {t/len(segs):8.1f} us")
PY
rm -rf $O/tr_$SH
cat $O/kernels_$SH.txt
