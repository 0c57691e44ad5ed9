set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wd_prof
rm -rf $O && mkdir -p $O
HEAD="--no-cpu-baseline --legs none"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wd -- python3 bench.py --workload wikidiverse --steps 10 --warmup 3 $HEAD > $O/wd_bench.json 2> $O/wd.err
echo "[wd] bf16x3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wd_if16 -- python3 bench.py --workload wikidiverse --precision bf16x3_if16 --steps 10 --warmup 3 $HEAD > $O/wd_if16_bench.json 2> $O/wd_if16.err
echo "[wd] if16 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wd_bf16f -- python3 bench.py --workload wikidiverse --features bf16 --steps 10 --warmup 3 $HEAD > $O/wd_bf16f_bench.json 2> $O/wd_bf16f.err
echo "[wd] bf16 features done"
find $O -name "*kernel_stats.csv" | head
