#!/bin/bash
# Counter passes for the stream-kernel instantiations furthest from the roof: bf16-stored features at the
# headline shape, pooled text at N = 11 (WikiDiverse-shaped), with the fp32 token instantiation beside them as the yardstick.
# FETCH_SIZE / WRITE_SIZE in separate passes (TCC slots), one SQ pass each; rocprofv3 with --pmc alone, program right after `--`.
#   bash tools/pmc_stream_r6.sh r6      -> gpurun_out/pmc_stream/{r6_hbm_traffic.json, r6_stream_sq.json, counters.txt}
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_stream
TAG=${1:-r6}
rm -rf $O && mkdir -p $O
export PYTHONUNBUFFERED=1
rocprofv3 -L > $O/counters.txt 2>&1 || echo "[pmc_stream] counter list failed"
HEAD="--steps 3 --warmup 1 --no-cpu-baseline --legs none"
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
run() {  # section, bench arguments
  local sec=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${sec}_fetch -- python3 bench.py "$@" $HEAD > $O/${sec}_fetch.log 2>&1 || { echo "[pmc_stream] $sec fetch FAILED"; return 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${sec}_write -- python3 bench.py "$@" $HEAD > $O/${sec}_write.log 2>&1 || { echo "[pmc_stream] $sec write FAILED"; return 1; }
  python3 tools/collect_pmc.py $O/${sec}_fetch $O/${sec}_write $O/${TAG}_hbm_traffic.json kernels_$sec "python3 bench.py $* $HEAD" >> $O/hbm_traffic.txt
  rocprofv3 --pmc $SQ --output-format csv -d $O/${sec}_sq -- python3 bench.py "$@" $HEAD > $O/${sec}_sq.log 2>&1 || echo "[pmc_stream] $sec SQ pass FAILED"
  rocprofv3 --pmc $SQ2 --output-format csv -d $O/${sec}_sq2 -- python3 bench.py "$@" $HEAD > $O/${sec}_sq2.log 2>&1 || echo "[pmc_stream] $sec SQ2 pass FAILED (counter names?)"
  echo "[pmc_stream] $sec done"
}
run bf16_features --features bf16 || exit 1
run wikidiverse --workload wikidiverse || exit 1
run wikidiverse_bf16 --workload wikidiverse --features bf16 || exit 1
run wikimel_f32
python3 tools/collect_stream_sq.py $O $O/${TAG}_stream_sq.json > $O/stream_sq_summary.txt 2>&1 || echo "[pmc_stream] SQ reduction failed"
cat $O/hbm_traffic.txt $O/stream_sq_summary.txt
