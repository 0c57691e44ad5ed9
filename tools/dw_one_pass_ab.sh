# The one-pass weight-gradient experiment: step time, gradient error, training trajectory.
# Writes gpurun_out/dw_one_pass.txt
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/dw_one_pass.txt
: > $O
echo "== gradients of one step against the fp64 oracle" >> $O
python3 tools/probes/dw_grad_error.py 2>/dev/null | grep -v amdgpu >> $O
for i in 1 2; do for P in 3 1; do for B in 64 512; do
  echo "== step time, weight-gradient passes $P, B = $B (run $i)" >> $O
  DRIN_DW_PASSES=$P python3 bench.py --mode train --batch $B --steps 30 --warmup 30 --no-cpu-baseline --legs none 2>> gpurun_out/dw_one_pass.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('ms_per_step', round(d['ms_per_step'],4), 'gemm_x3 ms/step', round(d.get('kernel_ms_per_step',{}).get('gemm_x3',0),4))
" >> $O
done; done; done
echo "[dw] step times done"
echo "== trajectory: 60 steps at T = 64, weight-gradient passes 1 (3 passes: profiles/r4_trajectory_T64.txt)" >> $O
DRIN_DW_PASSES=1 python3 tools/trajectory_run.py 60 64 bf16x3 2>/dev/null | grep -v amdgpu | awk 'NR % 10 == 1 || /steps at/' >> $O
echo "[dw] 60 steps done"
echo "== trajectory: 200 steps at T = 64, weight-gradient passes 1" >> $O
DRIN_DW_PASSES=1 python3 tools/trajectory_run.py 200 64 bf16x3 2>/dev/null | grep -v amdgpu | awk 'NR % 20 == 1 || /steps at/' >> $O
cat $O
