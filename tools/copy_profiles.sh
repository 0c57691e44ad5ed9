#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/round/ into the tracked profiles/ names of a round:
#   tools/copy_profiles.sh r3
set -e
R=${1:?round tag, e.g. r3}
O=gpurun_out/round
P=profiles
stats() { ls -t $O/$1/*/*_kernel_stats.csv | head -1; }
cp "$(stats wm)" $P/${R}_wikimel_b4096_kernel_stats.csv
cp "$(stats tab)" $P/${R}_table_cache_b4096_kernel_stats.csv
cp "$(stats train)" $P/${R}_train_b64_kernel_stats.csv
cp "$(stats train512)" $P/${R}_train_b512_kernel_stats.csv
cp "$(stats train_rccl)" $P/${R}_train_b64_rccl_world1_kernel_stats.csv
cp $O/wm_bench_under_rocprof.json $P/${R}_wikimel_b4096_bench_under_rocprof.json
cp $O/table_cache_bench_under_rocprof.json $P/${R}_table_cache_b4096_bench_under_rocprof.json
cp $O/train64_bench_under_rocprof.json $P/${R}_train_b64_bench_under_rocprof.json
cp $O/train512_bench_under_rocprof.json $P/${R}_train_b512_bench_under_rocprof.json
cp $O/train64_rccl_world1_bench_under_rocprof.json $P/${R}_train_b64_rccl_world1_bench_under_rocprof.json
cp $O/config5_1M_stream.json $P/${R}_config5_1M_mentions_streamed.json
# (a round profiled in two gpurun calls leaves the per-entity-cache sections in hbm_traffic_part2.json: joined here)
python3 - $O/hbm_traffic.json $O/hbm_traffic_part2.json $P/${R}_hbm_traffic.json <<'PY'
import json, os, sys
a = json.load(open(sys.argv[1]))
if os.path.exists(sys.argv[2]):
    b = json.load(open(sys.argv[2]))
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            a[k].update(v)
        elif k not in a:
            a[k] = v
json.dump(a, open(sys.argv[3], "w"), indent=1)
PY
cp gpurun_out/pmc_mfma/${R}_mfma_pmc.json $P/${R}_mfma_pmc.json
cp "$(stats bf16f)" $P/${R}_wikimel_b4096_bf16_features_kernel_stats.csv
cp $O/bf16f_bench_under_rocprof.json $P/${R}_wikimel_b4096_bf16_features_bench_under_rocprof.json
cp gpurun_out/pmc_mfma/summary.txt $P/${R}_mfma_pmc_summary.txt
cp "$(stats tabm)" $P/${R}_table_cache_mixed_f16_b4096_kernel_stats.csv
cp $O/table_cache_mixed_f16_bench_under_rocprof.json $P/${R}_table_cache_mixed_f16_b4096_bench_under_rocprof.json
cp "$(stats wd)" $P/${R}_wikidiverse_b16384_kernel_stats.csv
cp "$(stats if16)" $P/${R}_wikimel_b4096_mixed_bf16x3_if16_kernel_stats.csv
cp $O/wd_bench_under_rocprof.json $P/${R}_wikidiverse_b16384_bench_under_rocprof.json
cp $O/if16_bench_under_rocprof.json $P/${R}_wikimel_b4096_mixed_bf16x3_if16_bench_under_rocprof.json
# the default line three times (one JSON line each)
cat $O/wm_bench.json $O/wm_bench_1.json $O/wm_bench_2.json > $P/${R}_wikimel_b4096_bench_all_legs.json
cat $O/wm_bench_line.json $O/wm_bench_line_1.json $O/wm_bench_line_2.json > $P/${R}_wikimel_b4096_bench_stdout_lines.json
ls -la $P | grep ${R}_ | wc -l
# every figure DESIGN.md quotes from these files must agree with them to 3 % (tags: tools/check_design_numbers.py)
python3 tools/check_design_numbers.py DESIGN.md
