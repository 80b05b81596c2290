#!/bin/bash
# Three rocprofv3 runs of the default bench on the GPU box (kernel-trace stats, then the two HBM
# counter passes, each in a run of its own as the MI355X guide prescribes).  Usage:
#   gpurun -- 'bash tools/profile_bench.sh r01c'   then on the CPU side:
#   python tools/summarize_profile.py stats   gpurun_out/<tag>_stats/s_kernel_stats.csv profiles/<tag>_kernel_stats.csv
#   python tools/summarize_profile.py traffic gpurun_out/<tag>_fetch/f_counter_collection.csv \
#          gpurun_out/<tag>_write/w_counter_collection.csv profiles/<tag>_hbm_traffic.json
set -e
tag=${1:-prof}
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- \
    python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --score-k 0 --prior-samples 0 > $out/${tag}_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -o f -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --score-k 0 --prior-samples 0 > $out/${tag}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -o w -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --score-k 0 --prior-samples 0 > $out/${tag}_write.log 2>&1
grep -h '"metric"' $out/${tag}_stats.log | cut -c1-400
