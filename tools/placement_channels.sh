#!/bin/bash
# VERDICT r5 item 9, per-channel view: N fresh processes, each under `rocprofv3 --pmc TCC_REQ TCC_EA0_RDREQ` with JSON output (the CSV
# sums a counter over its instances; the JSON keeps the 16 channels x 8 XCDs apart), reduced on the box to one line per process:
# the stream kernel's duration and how evenly its L2 requests / memory-side reads spread over the 128 channel instances.
N=${1:-6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/placement_channels
rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1
for i in $(seq 1 $N); do
  rocprofv3 --pmc TCC_REQ TCC_EA0_RDREQ --output-format json -d $O/raw_$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs none > $O/run_$i.log 2>&1 || echo "[placement_channels] run $i FAILED"
  python3 tools/reduce_channel_json.py $O/raw_$i $O/channels_$i.json >> $O/table.txt 2>> $O/reduce.err || echo "[placement_channels] reduce $i failed"
  rm -rf $O/raw_$i
  echo "[placement_channels] $i done"
done
cat $O/table.txt; tail -5 $O/reduce.err
