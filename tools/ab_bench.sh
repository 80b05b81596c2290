# same-box A/B of the whole bench: an earlier tree unpacked under tools/ab/<name>/ (git archive
# <commit> tracevae_amd bench.py include | tar -x, library built in place) against this tree
cd $GRAFT_REPO_ROOT
old=${1:-tools/ab/r02_tree}
pick() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); s=d['scoring']; print('$1', 'ms/step', round(d['ms_per_step'],3), 'traces/s', round(d['value']), 'gemm TF', round(d['roofline']['achieved'],1), '| scoring ms', round(s['ms_per_step'],2), 'traces/s', round(s['value']), 'gemm TF', round(s['roofline']['achieved'],1))"; }
for i in 1 2; do
(cd $old && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --prior-samples 0 2>/dev/null) | pick old || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --prior-samples 0 2>/dev/null | pick new || exit 1
done
for b in 64 128; do
(cd $old && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --score-k 0 --prior-samples 0 --batch-size $b 2>/dev/null) | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('old batch', $b, round(d['ms_per_step'],3))" || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --score-k 0 --prior-samples 0 --batch-size $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('new batch', $b, round(d['ms_per_step'],3))" || exit 1
done
