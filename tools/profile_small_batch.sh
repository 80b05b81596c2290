#!/bin/bash
# kernel trace of the training step at a small batch (launch-bound regime): which launches make
# up a step, in order.  Usage: gpurun -- 'bash tools/profile_small_batch.sh r03_b64 64'
set -e
tag=${1:-b64}
bs=${2:-64}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- \
    python3 bench.py --steps 10 --warmup 3 --score-k 0 --batch-size $bs --no-cpu-baseline \
    --prior-samples 0 --pool 65536 --sweep-batches= > $out/${tag}_stats.log 2>&1
grep -h '"metric"' $out/${tag}_stats.log | cut -c1-300
