# same-box A/B of the whole step: tools/ab/libtv_old.so against the tree's library
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],3), 'gemm TF', round(d['roofline']['achieved'],1), 'scoring ms', round(d['scoring']['ms_per_step'],2), 'score gemm TF', round(d['scoring']['roofline']['achieved'],1))"; }
for i in 1 2; do
TRACEVAE_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libtv_old.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | pick old || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | pick new || exit 1
done
