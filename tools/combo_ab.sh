# bench.py's headline and configs under several knob settings: tools/combo_ab.sh <tag> "<tune A>" "<tune B>" ...
set -e
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for t in "$@"; do
  args=""; for kv in $(echo $t | tr ',' ' '); do args="$args --tune $kv"; done
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32 --no-pixel $args > $O/b_$t.json 2>/dev/null
  python - "$t" $O/b_$t.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print(sys.argv[1], round(d['ms_per_step'],4), round(d['passes']['min_ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in d['configs'].items()})
PY
done
