# What bounds k_cached_pairs besides HBM?  The same call over a table so small that every cache row is L2 / MALL resident:
# the kernel's time is then its VALU / LDS / L2 floor.  (One GPU call; writes gpurun_out/cached_floor.txt)
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cached_floor.txt
: > $O
for E in ${ENTS:-1000 8000 1000000}; do
  echo "== entities $E" >> $O
  python3 bench.py --workload table --batch 4096 --entity-cache --entities $E --steps 5 --warmup 2 --no-cpu-baseline --legs none 2>> gpurun_out/cached_floor.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
print('kernel_ms', d.get('kernel_ms_per_step'))
print('roofline', {k:d['roofline'].get(k) for k in ('kernel','avg_launch_ms','achieved','frac')})
" >> $O
done
cat $O
