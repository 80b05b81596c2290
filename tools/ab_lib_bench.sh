# same-box A/B of the whole bench between library builds with the same C-ABI: tools/ab/libtv_*.so
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); s=d['scoring']; print('$1', 'ms/step', round(d['ms_per_step'],3), 'gemm TF', round(d['roofline']['achieved'],1), '| scoring ms', round(s['ms_per_step'],2), 'gemm TF', round(s['roofline']['achieved'],1))"; }
for i in 1 2; do
for lib in tools/ab/libtv_*.so; do
TRACEVAE_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --prior-samples 0 2>/dev/null | pick $(basename $lib .so) || exit 1
done
done
