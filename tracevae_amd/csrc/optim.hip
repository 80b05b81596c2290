// Optimizer over one flat fp32 parameter buffer: L2 term / gradient, global gradient norm
// (two fixed-order stages, float64 partials), clip + RMSprop / Adam update in one pass.
// HBM-bound: the update reads p, g, state and writes p, state: 20 B (RMSprop) or 28 B (Adam)
// per parameter.
#include "tv_common.h"

namespace {

constexpr int RB_THREADS = 256;
constexpr int64_t RB_MAX_BLOCKS = 1024;

// sum of squares over the (masked) flat buffer: 16 bytes of x and 4 mask bytes per thread and trip,
// four fp32 accumulators folded into a double every 16 trips (the scalar form - one float and one
// mask byte per trip behind a single dependent fma chain - read the 360 MB of the configs[4]
// parameters at 1.1 TB/s)
template <bool VEC>
__global__ __launch_bounds__(RB_THREADS) void sumsq_partial_kernel(int64_t n,
                                                                   const float *__restrict__ x,
                                                                   const uint8_t *__restrict__ mask,
                                                                   double *__restrict__ partial) {
  __shared__ double red[RB_THREADS];
  const int64_t tid = (int64_t)blockIdx.x * RB_THREADS + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * RB_THREADS;
  double dacc = 0.0;
  const int64_t n4 = VEC ? (n >> 2) : 0;
  if (VEC) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int cnt = 0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const uint32_t *m4 = reinterpret_cast<const uint32_t *>(mask);
#pragma unroll 2
    for (int64_t i = tid; i < n4; i += stride) {
      float4 v = x4[i];
      if (mask != nullptr) {
        const uint32_t m = m4[i];
        if (!(m & 0x000000ffu)) v.x = 0.f;
        if (!(m & 0x0000ff00u)) v.y = 0.f;
        if (!(m & 0x00ff0000u)) v.z = 0.f;
        if (!(m & 0xff000000u)) v.w = 0.f;
      }
      a0 = fmaf(v.x, v.x, a0); a1 = fmaf(v.y, v.y, a1);
      a2 = fmaf(v.z, v.z, a2); a3 = fmaf(v.w, v.w, a3);
      if (++cnt == 16) { dacc += (double)((a0 + a1) + (a2 + a3)); a0 = a1 = a2 = a3 = 0.f; cnt = 0; }
    }
    dacc += (double)((a0 + a1) + (a2 + a3));
  }
  float acc = 0.f;
  int cnt = 0;
  for (int64_t i = 4 * n4 + tid; i < n; i += stride) {
    const float v = (mask == nullptr || mask[i]) ? x[i] : 0.f;
    acc = fmaf(v, v, acc);
    if (++cnt == 64) { dacc += (double)acc; acc = 0.f; cnt = 0; }
  }
  red[threadIdx.x] = dacc + (double)acc;
  __syncthreads();
  for (int s = RB_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(RB_THREADS) void sum_partials_kernel(int64_t nblk,
                                                                  const double *__restrict__ partial,
                                                                  int take_sqrt,
                                                                  float *__restrict__ out) {
  __shared__ double red[RB_THREADS];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < nblk; i += RB_THREADS) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = RB_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(take_sqrt ? sqrt(red[0]) : red[0]);
}

template <bool VEC>
__global__ __launch_bounds__(256) void axpy_masked_kernel(int64_t n, float alpha,
                                                          const float *__restrict__ x,
                                                          const uint8_t *__restrict__ mask,
                                                          float *__restrict__ y) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = VEC ? (n >> 2) : 0;
  if (VEC) {
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    const uint32_t *m4 = reinterpret_cast<const uint32_t *>(mask);
    for (int64_t i = tid; i < n4; i += stride) {
      const uint32_t m = mask ? m4[i] : 0x01010101u;
      if (m == 0) continue;                            // (y is not touched where the mask is off)
      const float4 xv = x4[i];
      float4 yv = y4[i];
      if (m & 0x000000ffu) yv.x = fmaf(alpha, xv.x, yv.x);
      if (m & 0x0000ff00u) yv.y = fmaf(alpha, xv.y, yv.y);
      if (m & 0x00ff0000u) yv.z = fmaf(alpha, xv.z, yv.z);
      if (m & 0xff000000u) yv.w = fmaf(alpha, xv.w, yv.w);
      y4[i] = yv;
    }
  }
  for (int64_t i = 4 * n4 + tid; i < n; i += stride)
    if (mask == nullptr || mask[i]) y[i] = fmaf(alpha, x[i], y[i]);
}

__global__ __launch_bounds__(256) void optim_step_kernel(int kind, int64_t n, float *__restrict__ p,
                                                         const float *__restrict__ g,
                                                         float *__restrict__ s1,
                                                         float *__restrict__ s2,
                                                         const float *__restrict__ grad_norm,
                                                         float clip_norm, float lr, float beta1,
                                                         float beta2, float eps, float bc1,
                                                         float bc2_sqrt,
                                                         const float *__restrict__ guard,
                                                         int guard_kind, float *guard_copy) {
  // the NaN decision of the step taken on the device (tv_optim_step_guarded): the same value
  // for every thread of the grid, so either all of them update or none does
  if (guard != nullptr) {
    const float v = guard[0];
    // the host's copy of the scalar (pinned host memory, mapped into the device's address
    // space): one 4-byte vector store instead of a device-to-host copy queued behind the step
    if (guard_copy != nullptr && blockIdx.x == 0 && threadIdx.x == 0) guard_copy[0] = v;
    if (guard_kind == 1 ? (v != v) : (v != 0.f)) return;
  }
  float coef = 1.0f;
  if (grad_norm != nullptr && clip_norm > 0.f) {
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    coef = fminf(clip_norm / (grad_norm[0] + 1e-6f), 1.0f);
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    if (kind == 0) {  // RMSprop: sq = a*sq + (1-a)*g^2 ; p -= lr * g / (sqrt(sq) + eps)
      const float sq = beta2 * s1[i] + (1.0f - beta2) * gi * gi;
      s1[i] = sq;
      p[i] -= lr * gi / (sqrtf(sq) + eps);
    } else {          // Adam with bias correction
      const float m = beta1 * s1[i] + (1.0f - beta1) * gi;
      const float v = beta2 * s2[i] + (1.0f - beta2) * gi * gi;
      s1[i] = m;
      s2[i] = v;
      p[i] -= (lr / bc1) * m / (sqrtf(v) / bc2_sqrt + eps);
    }
  }
}

inline int64_t reduce_blocks(int64_t n) {
  int64_t b = tv_cdiv(n, (int64_t)RB_THREADS * 16);
  if (b < 1) b = 1;
  if (b > RB_MAX_BLOCKS) b = RB_MAX_BLOCKS;
  return b;
}

inline unsigned ew_blocks(int64_t n) {
  int64_t b = tv_cdiv(n, 256 * 4);
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (unsigned)b;
}

}  // namespace

extern "C" int64_t tv_reduce_blocks(int64_t n) { return reduce_blocks(n); }

extern "C" int tv_sumsq_partial(int64_t n, const float *x, const uint8_t *mask, double *partial,
                                void *stream) {
  TV_REQUIRE(n >= 1 && x && partial, "tv_sumsq_partial: bad arguments");
  if ((((uintptr_t)x & 15) | ((uintptr_t)mask & 3)) == 0)
    hipLaunchKernelGGL(sumsq_partial_kernel<true>, dim3((unsigned)reduce_blocks(n)), dim3(RB_THREADS), 0,
                       (hipStream_t)stream, n, x, mask, partial);
  else
    hipLaunchKernelGGL(sumsq_partial_kernel<false>, dim3((unsigned)reduce_blocks(n)), dim3(RB_THREADS), 0,
                       (hipStream_t)stream, n, x, mask, partial);
  TV_LAUNCH_CHECK("tv_sumsq_partial");
  return TV_OK;
}

extern "C" int tv_sum_partials(int64_t nblk, const double *partial, int take_sqrt, float *out,
                               void *stream) {
  TV_REQUIRE(nblk >= 1 && partial && out, "tv_sum_partials: bad arguments");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(RB_THREADS), 0, (hipStream_t)stream, nblk,
                     partial, take_sqrt, out);
  TV_LAUNCH_CHECK("tv_sum_partials");
  return TV_OK;
}

extern "C" int tv_axpy_masked(int64_t n, float alpha, const float *x, const uint8_t *mask, float *y,
                              void *stream) {
  TV_REQUIRE(n >= 1 && x && y, "tv_axpy_masked: bad arguments");
  if ((((uintptr_t)x | (uintptr_t)y) & 15) == 0 && ((uintptr_t)mask & 3) == 0)
    hipLaunchKernelGGL(axpy_masked_kernel<true>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, n,
                       alpha, x, mask, y);
  else
    hipLaunchKernelGGL(axpy_masked_kernel<false>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, n,
                       alpha, x, mask, y);
  TV_LAUNCH_CHECK("tv_axpy_masked");
  return TV_OK;
}

static int optim_step(const char *who, int kind, int64_t n, float *param, const float *grad,
                      float *state1, float *state2, const float *grad_norm, float clip_norm,
                      float lr, float beta1, float beta2, float eps, int64_t step,
                      const float *guard, int guard_kind, float *guard_host, void *stream) {
  TV_REQUIRE(kind == 0 || kind == 1, "tv_optim_step: kind must be 0 (RMSprop) or 1 (Adam)");
  TV_REQUIRE(n >= 1 && param && grad && state1, "tv_optim_step: bad arguments");
  TV_REQUIRE(kind == 0 || state2, "tv_optim_step: Adam needs two state buffers");
  TV_REQUIRE(step >= 1, "tv_optim_step: step counts from 1");
  TV_REQUIRE(guard_kind >= 0 && guard_kind <= 2 && (guard != nullptr) == (guard_kind != 0),
             "tv_optim_step_guarded: guard_kind 1 (NaN) or 2 (non-zero) with a guard, 0 without");
  TV_REQUIRE(!guard_host || guard, "tv_optim_step_guarded: guard_host without a guard");
  float *guard_copy = nullptr;
  if (guard_host) {
    // the device's address of the pinned host scalar; fails (instead of faulting in the kernel)
    // for memory that is not mapped
    void *mapped = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&mapped, guard_host, 0);
    if (e != hipSuccess || !mapped) {
      (void)hipGetLastError();   // the refusal is ours to report: leave no sticky HIP error behind
      tv_set_error("tv_optim_step_guarded: guard_host is not pinned host memory (%s)",
                   hipGetErrorString(e));
      return TV_ERR_INVALID;
    }
    guard_copy = (float *)mapped;
  }
  float bc1 = 1.0f, bc2s = 1.0f;
  if (kind == 1) {
    bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  }
  hipLaunchKernelGGL(optim_step_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, kind,
                     n, param, grad, state1, state2, grad_norm, clip_norm, lr, beta1, beta2, eps,
                     bc1, bc2s, guard, guard_kind, guard_copy);
  TV_LAUNCH_CHECK(who);
  return TV_OK;
}

extern "C" int tv_optim_step(int kind, int64_t n, float *param, const float *grad, float *state1,
                             float *state2, const float *grad_norm, float clip_norm, float lr,
                             float beta1, float beta2, float eps, int64_t step, void *stream) {
  return optim_step("tv_optim_step", kind, n, param, grad, state1, state2, grad_norm, clip_norm, lr,
                    beta1, beta2, eps, step, nullptr, 0, nullptr, stream);
}

extern "C" int tv_optim_step_guarded(int kind, int64_t n, float *param, const float *grad,
                                     float *state1, float *state2, const float *grad_norm,
                                     float clip_norm, float lr, float beta1, float beta2, float eps,
                                     int64_t step, const float *guard, int guard_kind,
                                     float *guard_host, void *stream) {
  return optim_step("tv_optim_step_guarded", kind, n, param, grad, state1, state2, grad_norm,
                    clip_norm, lr, beta1, beta2, eps, step, guard, guard_kind, guard_host, stream);
}
