#!/bin/bash
# Regenerates the judged evidence under gpurun_out/round/ (copy the summaries into profiles/ afterwards):
#   kernel-trace stats + the bench line of the same command (default workload), the full default bench line (all legs),
#   PMC traffic passes (their own runs), cached-table and training-step kernel stats; the SQ / MFMA counter passes are
#   tools/pmc_mfma.sh (separate call: it writes gpurun_out/pmc_mfma/).
# rocprofv3: program directly after `--`, counters in their own passes.
# The whole script takes ~25 minutes - more than one gpurun call allows: `tools/profile_round.sh r6 1` (headline, PMC, bf16 features,
# WikiDiverse, fp16 contraction) and `tools/profile_round.sh r6 2` (per-entity cache, training, config 5, two more default lines) are
# two calls; part 2 keeps what part 1 left under gpurun_out/round/ (the directory is merged back between calls and travels with the tree).
set -e
TAG=${1:-r6}
PART=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round
if [ "$PART" != "2" ]; then rm -rf $O; fi
mkdir -p $O
export PYTHONUNBUFFERED=1
HEAD="--no-cpu-baseline --legs none"
if [ "$PART" != "2" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wm -- python3 bench.py --steps 10 --warmup 3 $HEAD --legs-file $O/wm_bench_under_rocprof.json > /dev/null 2> $O/wm.err
echo "[profile_round] default kernel stats done"
python3 bench.py --legs-file $O/wm_bench.json > $O/wm_bench_line.json 2> $O/wm_bench.err
echo "[profile_round] full default line done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 $HEAD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 $HEAD > $O/pmc_write.log 2>&1
python3 tools/collect_pmc.py $O/pmc_fetch $O/pmc_write $O/hbm_traffic.json kernels > $O/hbm_traffic.txt
echo "[profile_round] default PMC done"
# the bf16-stored-features leg: kernel stats of its own command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16f -- python3 bench.py --features bf16 --steps 10 --warmup 3 $HEAD --legs-file $O/bf16f_bench_under_rocprof.json > /dev/null 2> $O/bf16f.err
echo "[profile_round] mixed precision / bf16 features done"
# WikiDiverse-shaped (BASELINE config 2): kernel stats of the default arithmetic (the one-pass image contraction is gated on N >= 64)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wd -- python3 bench.py --workload wikidiverse --steps 10 --warmup 3 $HEAD --legs-file $O/wd_bench_under_rocprof.json > /dev/null 2> $O/wd.err
# the fp16 image contraction at the headline's shape (the mode's own kernel stats)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/if16 -- python3 bench.py --precision bf16x3_if16 --steps 10 --warmup 3 $HEAD --legs-file $O/if16_bench_under_rocprof.json > /dev/null 2> $O/if16.err
echo "[profile_round] wikidiverse / fp16 contraction done"
fi
if [ "$PART" = "1" ]; then ls $O; exit 0; fi
# (gpurun_out/ does not travel to the GPU box: part 2 starts from an empty directory there, and what it writes is merged over the
#  local copy afterwards - so its traffic sections go to their own file; tools/copy_profiles.sh joins the two)
HT=$O/hbm_traffic.json
if [ "$PART" = "2" ]; then HT=$O/hbm_traffic_part2.json; fi
TAB="--workload table --batch 4096 --entity-cache"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tab -- python3 bench.py $TAB --steps 5 --warmup 2 $HEAD --legs-file $O/table_cache_bench_under_rocprof.json > /dev/null 2> $O/tab.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_tab -- python3 bench.py $TAB --steps 2 --warmup 1 $HEAD > $O/pmc_fetch_tab.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_tab -- python3 bench.py $TAB --steps 2 --warmup 1 $HEAD > $O/pmc_write_tab.log 2>&1
python3 tools/collect_pmc.py $O/pmc_fetch_tab $O/pmc_write_tab $HT kernels_table_cache "python3 bench.py $TAB --steps 2 --warmup 1 $HEAD" >> $O/hbm_traffic.txt
# the same chunk from DRIN_CACHE_MIXED_F16 rows: kernel stats + its own PMC passes (section kernels_table_cache_mixed_f16)
TABM="$TAB --cache-format mixed_f16"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tabm -- python3 bench.py $TABM --steps 5 --warmup 2 $HEAD --legs-file $O/table_cache_mixed_f16_bench_under_rocprof.json > /dev/null 2> $O/tabm.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_tabm -- python3 bench.py $TABM --steps 2 --warmup 1 $HEAD > $O/pmc_fetch_tabm.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_tabm -- python3 bench.py $TABM --steps 2 --warmup 1 $HEAD > $O/pmc_write_tabm.log 2>&1
python3 tools/collect_pmc.py $O/pmc_fetch_tabm $O/pmc_write_tabm $HT kernels_table_cache_mixed_f16 "python3 bench.py $TABM --steps 2 --warmup 1 $HEAD" >> $O/hbm_traffic.txt
echo "[profile_round] table cache done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 bench.py --mode train --batch 64 --steps 20 --warmup 30 --legs-file $O/train64_bench_under_rocprof.json > /dev/null 2> $O/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train512 -- python3 bench.py --mode train --batch 512 --steps 20 --warmup 30 --legs-file $O/train512_bench_under_rocprof.json > /dev/null 2> $O/train512.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_rccl -- python3 bench.py --mode train --batch 64 --steps 20 --warmup 30 --force-collective --legs-file $O/train64_rccl_world1_bench_under_rocprof.json > /dev/null 2> $O/train_rccl.err
echo "[profile_round] training done"
python3 bench.py --workload table --entity-cache --mentions 1000000 --chunk 4096 --legs-file $O/config5_1M_stream.json > /dev/null 2> $O/config5.err
echo "[profile_round] config 5 full stream done"
for i in 1 2; do python3 bench.py --legs-file $O/wm_bench_$i.json > $O/wm_bench_line_$i.json 2>> $O/wm_bench.err; done
echo "[profile_round] two more default lines done"
# (the SQ / MFMA counter passes are their own gpurun call: bash tools/pmc_mfma.sh $TAG, which leaves gpurun_out/pmc_mfma/${TAG}_mfma_pmc.json
#  and summary.txt; tools/copy_profiles.sh picks them up from there)
ls $O
