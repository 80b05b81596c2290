# A/B of the GEMM: tools/ab/libtv_old.so (a build of an earlier gemm_f32.hip) against the tree's library
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q -k "gemm" --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
for w in fwd dx dw fwdact dwact; do
  TRACEVAE_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libtv_old.so timeout -k 10 60 python tools/bench_gemm_one.py $w 100 2>/dev/null | sed 's/^/old /' || exit 1
  timeout -k 10 60 python tools/bench_gemm_one.py $w 100 2>/dev/null | sed 's/^/new /' || exit 1
done
