# same-box A/B of the long-trace preset (BASELINE configs[4]) between library builds: tools/ab/libtv_*.so
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for lib in tools/ab/libtv_*.so; do
TRACEVAE_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --batch-size 32 --max-node-count 500 --num-operations 10001 --gnn-width 1000 --z-dim 20 --score-k 0 --prior-samples 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$(basename $lib .so)', 'c5 ms/step', round(d['ms_per_step'],3), 'gemm TF', round(d['roofline']['achieved'],1))" || exit 1
done
done
