#!/bin/bash
# Regenerates the judged evidence under gpurun_out/ (copy the summaries into profiles/ afterwards):
#   kernel-trace stats + the bench line of the same command (default workload), PMC traffic passes, cached-table and
#   training-step kernel stats.  rocprofv3: program directly after `--`, counters in their own passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wm -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/wm_bench_under_rocprof.json 2> $O/wm.err
python3 bench.py > $O/wm_bench.json 2> $O/wm_bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
python3 tools/collect_pmc.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.json > $O/hbm_traffic.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tab -- python3 bench.py --workload table --batch 256 --entity-cache --steps 10 --warmup 3 --no-cpu-baseline > $O/table_cache_bench_under_rocprof.json 2> $O/tab.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 bench.py --mode train --batch 64 --steps 20 --warmup 3 > $O/train64_bench_under_rocprof.json 2> $O/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train512 -- python3 bench.py --mode train --batch 512 --steps 20 --warmup 30 > $O/train512_bench_under_rocprof.json 2> $O/train512.err
python3 bench.py --mode train --batch 512 --train-form table > $O/train512_table_bench.json 2> $O/train512_table.err
python3 bench.py --workload wikidiverse --no-cpu-baseline > $O/wd_bench.json 2> $O/wd.err
python3 bench.py --features bf16 --no-cpu-baseline > $O/wm_bf16_features_bench.json 2> $O/wmbf.err
ls $O
