# A/B of the GEMM: tools/ab/libtv_old.so (a build of an earlier gemm_f32.hip, same C-ABI) against the tree's library
cd $GRAFT_REPO_ROOT
for w in fwd dx dw fwdact dwact; do
  TRACEVAE_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libtv_old.so timeout -k 10 60 python tools/bench_gemm_one.py $w 100 2>/dev/null | sed 's/^/old /' || exit 1
  timeout -k 10 60 python tools/bench_gemm_one.py $w 100 2>/dev/null | sed 's/^/new /' || exit 1
done
