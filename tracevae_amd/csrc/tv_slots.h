// Second stage of the two-stage column reductions: sum `nblk` partial slots per column.
// A 256-thread workgroup handles TV_SLOT_COLS columns; each column is summed by
// TV_SLOT_GROUPS threads that walk the slots with a fixed stride (slot b goes to group
// b % TV_SLOT_GROUPS), then the groups are combined in group order through LDS.  The
// order is fixed, so the result is bitwise reproducible; accumulation is float64.
#pragma once
#include "tv_common.h"

#define TV_SLOT_COLS 8
#define TV_SLOT_GROUPS 32

static inline unsigned tv_slot_grid(int64_t n) { return (unsigned)tv_cdiv(n, TV_SLOT_COLS); }

// Sums partial[b * slot_stride + idx] over b for NS consecutive "planes" of a slot
// (plane p lives at offset p * plane_stride inside the slot).  Every thread of the
// workgroup must call it; the totals are valid in the threads of group 0.
template <int NS>
__device__ __forceinline__ void tv_slot_sums(const float *__restrict__ partial, int64_t nblk,
                                             int64_t slot_stride, int64_t plane_stride,
                                             int64_t idx, bool valid, double (&total)[NS]) {
  __shared__ double lds[NS][TV_SLOT_GROUPS][TV_SLOT_COLS];
  const int cl = threadIdx.x % TV_SLOT_COLS, grp = threadIdx.x / TV_SLOT_COLS;
  double acc[NS];
#pragma unroll
  for (int p = 0; p < NS; ++p) acc[p] = 0.0;
  if (valid) {
    // 4 slots per trip so the loads of a trip are in flight together; the additions stay in
    // slot order
    int64_t b = grp;
    for (; b + 3 * TV_SLOT_GROUPS < nblk; b += 4 * TV_SLOT_GROUPS) {
      float v[4][NS];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < NS; ++p)
          v[j][p] = partial[(b + j * TV_SLOT_GROUPS) * slot_stride + p * plane_stride + idx];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < NS; ++p) acc[p] += (double)v[j][p];
    }
    for (; b < nblk; b += TV_SLOT_GROUPS) {
#pragma unroll
      for (int p = 0; p < NS; ++p) acc[p] += (double)partial[b * slot_stride + p * plane_stride + idx];
    }
  }
#pragma unroll
  for (int p = 0; p < NS; ++p) lds[p][grp][cl] = acc[p];
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    double t = 0.0;
    if (grp == 0)
      for (int g = 0; g < TV_SLOT_GROUPS; ++g) t += lds[p][g][cl];
    total[p] = t;
  }
}
