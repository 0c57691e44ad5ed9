# Config 5 from fp32 cache rows and from DRIN_CACHE_MIXED_F16 rows, same box, alternating (writes gpurun_out/mixed_cache_ab.txt)
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/mixed_cache_ab.txt
: > $O
for i in 1 2; do
for F in f32 mixed_f16; do
  echo "== cache format $F (run $i)" >> $O
  python3 bench.py --workload table --batch 4096 --entity-cache --cache-format $F --steps 5 --warmup 2 --no-cpu-baseline --legs none 2>> gpurun_out/mixed_cache_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('ms_per_step', round(d['ms_per_step'],3), 'M pairs/s', round(d['value']/1e6,2), d['path'])
print('kernel_ms', {k:round(v,3) for k,v in d.get('kernel_ms_per_step',{}).items() if v})
print('roofline', {k:d['roofline'].get(k) for k in ('kernel','avg_launch_ms','achieved','frac','algorithmic_bytes_per_launch')})
" >> $O
done
done
cat $O
