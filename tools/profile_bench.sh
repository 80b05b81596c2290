#!/bin/bash
# rocprofv3 runs of the default bench on the GPU box: kernel-trace stats, then the two HBM counter
# passes (each in a run of its own, as the MI355X guide prescribes), for the training leg and for
# the K = 100 scoring leg.  Usage:
#   gpurun -- 'bash tools/profile_bench.sh r03'   then on the CPU side: bash tools/summarize_all.sh r03
set -e
tag=${1:-prof}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
small="--no-cpu-baseline --prior-samples 0 --pool 65536 --sweep-batches="
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- \
    python3 bench.py --steps 5 --warmup 2 --score-k 0 $small > $out/${tag}_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -o f -- \
    python3 bench.py --steps 2 --warmup 1 --score-k 0 $small > $out/${tag}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -o w -- \
    python3 bench.py --steps 2 --warmup 1 --score-k 0 $small > $out/${tag}_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_score -o s -- \
    python3 bench.py --scoring-only --score-steps 3 $small > $out/${tag}_score.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_sfetch -o f -- \
    python3 bench.py --scoring-only --score-steps 1 $small > $out/${tag}_sfetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_swrite -o w -- \
    python3 bench.py --scoring-only --score-steps 1 $small > $out/${tag}_swrite.log 2>&1
grep -h '"metric"' $out/${tag}_stats.log | cut -c1-400
