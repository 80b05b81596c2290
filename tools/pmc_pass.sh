#!/bin/bash
# one rocprofv3 counter pass over a short bench leg.  Usage:
#   gpurun -- 'bash tools/pmc_pass.sh <tag> "<counters>" [bench flags...]'
set -e
tag=$1; ctr=$2; shift 2
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/${tag} -o p -- \
    python3 bench.py --no-cpu-baseline --prior-samples 0 --pool 65536 --sweep-batches= "$@" > $out/${tag}.log 2>&1
