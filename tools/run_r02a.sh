set -e
export TMPDIR=/tmp
out=gpurun_out
timeout -k 10 280 python3 bench.py --steps 10 --warmup 3 --gemm-shapes > $out/r02a_bench.log 2> $out/r02a_shapes.log
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r02a_c5 -o s -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --score-k 0 --prior-samples 0 --batch-size 8 --max-node-count 500 --num-operations 10001 --gnn-width 1000 --z-dim 20 > $out/r02a_c5.log 2>&1
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r02a_stats -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --score-k 0 --prior-samples 0 > $out/r02a_stats.log 2>&1
cut -c1-600 $out/r02a_bench.log
