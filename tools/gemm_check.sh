# same-box A/B of GEMM builds: every tools/ab/libtv_*.so (builds of other gemm_f32.hip variants, same C-ABI)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for w in ${GEMM_CASES:-fwd dx dw fwdact dwact}; do
  for lib in tools/ab/libtv_*.so; do
    TRACEVAE_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 60 python tools/bench_gemm_one.py $w 100 2>/dev/null | sed "s|^|$(basename $lib .so) |" || exit 1
  done
done
done
