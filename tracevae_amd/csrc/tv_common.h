// Shared helpers of the gfx950 kernels (wave64, CDNA4).  Not a compatibility layer: this
// code is written for MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/tracevae_hip.h"

#define TV_WAVE 64

void tv_set_error(const char *fmt, ...);

#define TV_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      tv_set_error(__VA_ARGS__);         \
      return TV_ERR_INVALID;             \
    }                                    \
  } while (0)

#define TV_LAUNCH_CHECK(name)                                          \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      tv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return TV_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)

#define TV_HIP(call)                                                   \
  do {                                                                 \
    hipError_t e__ = (call);                                           \
    if (e__ != hipSuccess) {                                           \
      tv_set_error("%s failed: %s", #call, hipGetErrorString(e__));    \
      return TV_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)

static inline int64_t tv_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// number of partial slots of the two-stage column reductions (same rule everywhere)
// (workgroups of TV_STAT_WAVES waves, each wave walks rows with a fixed stride, the waves
// of a workgroup are combined in wave order, the workgroups in workgroup order)
#define TV_STAT_MAX_BLOCKS 512
#define TV_STAT_WAVES 8
static inline int64_t tv_stat_max_blocks(void) { return TV_STAT_MAX_BLOCKS; }   // two per CU
static inline int64_t tv_stat_blocks_host(int64_t rows) {
  int64_t b = tv_cdiv(rows, TV_STAT_WAVES);
  if (b < 1) b = 1;
  if (b > tv_stat_max_blocks()) b = tv_stat_max_blocks();
  return b;
}

__device__ __forceinline__ float tv_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float tv_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double tv_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float tv_lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

// softplus(x) = log(1 + exp(x)), stable
__device__ __forceinline__ float tv_softplus(float x) {
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}
