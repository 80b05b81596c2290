# the numbers DESIGN.md quotes, on one box: default bench, long-trace preset, scoring kernel table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-final}
out=gpurun_out
timeout -k 10 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --batch-size 32 --max-node-count 500 --num-operations 10001 --gnn-width 1000 --z-dim 20 --score-k 10 --score-batch 32 --prior-samples 0 > $out/${tag}_c5.json 2> $out/${tag}_c5.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_score -o s -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --prior-samples 0 --score-steps 3 > $out/${tag}_score.log 2>&1 || exit 1
echo done
