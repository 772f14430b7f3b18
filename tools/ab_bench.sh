# A/B of bench.py under different --tune settings, two rounds each (run on the GPU box).
# usage: tools/_ab.sh "<tune args A>" "<tune args B>" ...   (each a string of --tune flags; "" = defaults)
for rep in 1 2; do
for t in "$@"; do
  python bench.py --no-cpu-baseline $t > gpurun_out/bench_t.log 2>&1
  python - "$t" <<'PY'
import json,sys
l=open('gpurun_out/bench_t.log').read().strip().splitlines()[-1]
try:
    d=json.loads(l); r=d.get('roofline') or {}
    print(f"{sys.argv[1]!r:28} {d['value']/1e6:7.2f} M  upd {d['ms_per_step']:.3f} ms  roof {r.get('achieved',0):.0f} {r.get('unit')} frac {r.get('frac',0):.3f} avg {r.get('avg_launch_ms',0)*1e3:.1f} us x{r.get('launches')}", flush=True)
except Exception as e:
    print(sys.argv[1], 'ERR', l[-300:])
PY
done; done
