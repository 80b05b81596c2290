// Skinny dense helpers that are HBM-bound rather than MFMA-bound: a right-hand side with
// at most 8 columns (the 1000->2 latency head) and column sums (bias gradients).
// Fixed-order two-stage reductions, no atomics.
#include "tv_common.h"
#include "tv_slots.h"

namespace {

constexpr int MAXN = 8;

// wave per row: out[r,j] = scale[r] * sum_k x[r,k] w[k,j]
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(int64_t rows, int64_t K, int N,
                                                         const float *__restrict__ x,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ scale,
                                                         float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = wave; r < rows; r += nwaves) {
    float acc[MAXN];
#pragma unroll
    for (int j = 0; j < MAXN; ++j) acc[j] = 0.f;
    for (int64_t k = lane; k < K; k += 64) {
      const float v = x[r * K + k];
#pragma unroll
      for (int j = 0; j < MAXN; ++j)
        if (j < N) acc[j] = fmaf(v, w[k * N + j], acc[j]);
    }
    const float sc = scale ? scale[r] : 1.0f;
#pragma unroll
    for (int j = 0; j < MAXN; ++j) {
      if (j < N) {
        const float t = tv_wave_sum(acc[j]);
        if (lane == 0) out[r * N + j] = t * sc;
      }
    }
  }
}

// dx[r,k] = scale[r] * sum_j dout[r,j] w[k,j]
__global__ __launch_bounds__(256) void rowdot_bwd_dx_kernel(int64_t rows, int64_t K, int N,
                                                            const float *__restrict__ w,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ dout,
                                                            float *__restrict__ dx) {
  const int64_t total = rows * K, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / K, k = e - r * K;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc = fmaf(dout[r * N + j], w[k * N + j], acc);
    dx[e] = acc * (scale ? scale[r] : 1.0f);
  }
}

// partial[by][k*N+j] = sum over this workgroup's rows of scale[r] x[r,k] dout[r,j]
__global__ __launch_bounds__(256) void rowdot_bwd_dw_kernel(int64_t rows, int64_t K, int N,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ dout,
                                                            float *__restrict__ partial) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nblk = gridDim.y, by = blockIdx.y;
  if (k >= K) return;
  float acc[MAXN];
#pragma unroll
  for (int j = 0; j < MAXN; ++j) acc[j] = 0.f;
  for (int64_t r = by; r < rows; r += nblk) {
    const float v = x[r * K + k] * (scale ? scale[r] : 1.0f);
#pragma unroll
    for (int j = 0; j < MAXN; ++j)
      if (j < N) acc[j] = fmaf(v, dout[r * N + j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < MAXN; ++j)
    if (j < N) partial[by * K * N + k * N + j] = acc[j];
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(int64_t rows, int64_t C, int64_t ld,
                                                             const float *__restrict__ x,
                                                             float *__restrict__ partial) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nblk = gridDim.y, by = blockIdx.y;
  if (c >= C) return;
  float acc = 0.f;
  for (int64_t r = by; r < rows; r += nblk) acc += x[r * ld + c];
  partial[by * C + c] = acc;
}

__global__ __launch_bounds__(256) void sum_slots_kernel(int64_t nblk, int64_t n,
                                                        const float *__restrict__ partial,
                                                        float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * TV_SLOT_COLS + threadIdx.x % TV_SLOT_COLS;
  double tot[1];
  tv_slot_sums<1>(partial, nblk, n, 0, i, i < n, tot);
  if (threadIdx.x < TV_SLOT_COLS && i < n) out[i] = (float)tot[0];
}

}  // namespace

extern "C" int tv_rowdot_fwd(int64_t rows, int64_t k, int64_t n, const float *x, const float *w,
                             const float *row_scale, float *out, void *stream) {
  TV_REQUIRE(rows >= 0 && k >= 1, "tv_rowdot_fwd: bad sizes");
  TV_REQUIRE(n >= 1 && n <= MAXN, "tv_rowdot_fwd: n must be in [1,%d]: got %lld", MAXN, (long long)n);
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && w && out, "tv_rowdot_fwd: null pointer");
  int64_t blocks = tv_cdiv(rows, 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     rows, k, (int)n, x, w, row_scale, out);
  TV_LAUNCH_CHECK("tv_rowdot_fwd");
  return TV_OK;
}

extern "C" int64_t tv_rowdot_bwd_workspace_bytes(int64_t rows, int64_t k, int64_t n) {
  return (int64_t)sizeof(float) * tv_stat_blocks_host(rows) * k * n;
}

extern "C" int tv_rowdot_bwd(int64_t rows, int64_t k, int64_t n, const float *x, const float *w,
                             const float *row_scale, const float *dout, float *dx, float *dw,
                             void *workspace, int64_t workspace_bytes, void *stream) {
  TV_REQUIRE(rows >= 1 && k >= 1, "tv_rowdot_bwd: bad sizes");
  TV_REQUIRE(n >= 1 && n <= MAXN, "tv_rowdot_bwd: n must be in [1,%d]: got %lld", MAXN, (long long)n);
  TV_REQUIRE(x && w && dout && dw, "tv_rowdot_bwd: null pointer");
  const int64_t need = tv_rowdot_bwd_workspace_bytes(rows, k, n);
  if (!workspace || workspace_bytes < need) {
    tv_set_error("tv_rowdot_bwd: workspace of %lld bytes needed, %lld given", (long long)need,
                 (long long)workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    int64_t blocks = tv_cdiv(rows * k, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rowdot_bwd_dx_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rows, k, (int)n,
                       w, row_scale, dout, dx);
    TV_LAUNCH_CHECK("tv_rowdot_bwd(dx)");
  }
  const int64_t nblk = tv_stat_blocks_host(rows);
  hipLaunchKernelGGL(rowdot_bwd_dw_kernel, dim3((unsigned)tv_cdiv(k, 256), (unsigned)nblk), dim3(256),
                     0, s, rows, k, (int)n, x, row_scale, dout, (float *)workspace);
  TV_LAUNCH_CHECK("tv_rowdot_bwd(dw)");
  hipLaunchKernelGGL(sum_slots_kernel, dim3(tv_slot_grid(k * n)), dim3(256), 0, s, nblk,
                     k * n, (const float *)workspace, dw);
  TV_LAUNCH_CHECK("tv_rowdot_bwd(reduce)");
  return TV_OK;
}

extern "C" int64_t tv_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  return (int64_t)sizeof(float) * tv_stat_blocks_host(rows) * cols;
}

extern "C" int tv_colsum(int64_t rows, int64_t cols, const float *x, int64_t ldx, float *out,
                         void *workspace, int64_t workspace_bytes, void *stream) {
  TV_REQUIRE(rows >= 1 && cols >= 1 && ldx >= cols, "tv_colsum: bad sizes");
  TV_REQUIRE(x && out, "tv_colsum: null pointer");
  const int64_t need = tv_colsum_workspace_bytes(rows, cols);
  if (!workspace || workspace_bytes < need) {
    tv_set_error("tv_colsum: workspace of %lld bytes needed, %lld given", (long long)need,
                 (long long)workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t nblk = tv_stat_blocks_host(rows);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)tv_cdiv(cols, 256), (unsigned)nblk),
                     dim3(256), 0, s, rows, cols, ldx, x, (float *)workspace);
  TV_LAUNCH_CHECK("tv_colsum(partial)");
  hipLaunchKernelGGL(sum_slots_kernel, dim3(tv_slot_grid(cols)), dim3(256), 0, s, nblk,
                     cols, (const float *)workspace, out);
  TV_LAUNCH_CHECK("tv_colsum(reduce)");
  return TV_OK;
}
