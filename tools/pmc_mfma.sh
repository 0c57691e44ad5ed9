#!/bin/bash
# SQ counter passes of the MFMA-bound configurations: headline, WikiDiverse-shaped scoring, the training
# step at batch 64 and 512.  One pass each (8 SQ counters + 1 GRBM); rocprofv3 with --pmc alone, the program right after `--`.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_mfma
TAG=${1:-r6}
rm -rf $O && mkdir -p $O
CNT="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
run() {  # section, bench arguments
  local sec=$1; shift
  rocprofv3 --pmc $CNT --output-format csv -d $O/$sec -- python3 bench.py "$@" > $O/$sec.log 2>&1
  python3 tools/collect_mfma_pmc.py $O/$sec $O/${TAG}_mfma_pmc.json $sec "python3 bench.py $*" >> $O/summary.txt
  echo "[pmc_mfma] $sec done"
}
run wikimel_b4096 --steps 3 --warmup 1 --no-cpu-baseline --legs none
run wikimel_b4096_if16 --precision bf16x3_if16 --steps 3 --warmup 1 --no-cpu-baseline --legs none
run wikidiverse_b16384 --workload wikidiverse --steps 3 --warmup 1 --no-cpu-baseline --legs none
run train_b64 --mode train --batch 64 --steps 5 --warmup 5
run train_b512 --mode train --batch 512 --steps 3 --warmup 3
cat $O/summary.txt
