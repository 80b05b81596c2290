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

// ------------------------------------------------------------------------------------
// Linear layers with few outputs (the z / z2 heads: out_features = 10): y = x W^T + b with W
// stored [N, K] like torch / tensorkit Linear, N <= 16.  At 512 rows these are far too small
// for the 128x128 MFMA tiles (a handful of workgroups walking k alone on their CUs); here a
// wave owns a row (forward) or a thread owns a weight column (gradients).
// ------------------------------------------------------------------------------------
constexpr int SKN = 16;

// The three kernels serve both orientations of a Linear with one small dimension (j < N <= 16)
// and one large (k < K); W(j,k) sits at w[j*wsj + k*wsk]:
//   few outputs (N = out, K = in,  W [N,K]): wsj = K, wsk = 1
//   few inputs  (N = in,  K = out, W [K,N]): wsj = 1, wsk = N

// wave per row: out[r,j] = sum_k a[r,k] W(j,k) (+ bias[j])
__global__ __launch_bounds__(256) void skinny_rowdot_kernel(int64_t rows, int64_t K, int N,
                                                            int64_t wsj, int64_t wsk,
                                                            const float *__restrict__ a,
                                                            const float *__restrict__ w,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = wave; r < rows; r += nwaves) {
    float acc[SKN];
#pragma unroll
    for (int j = 0; j < SKN; ++j) acc[j] = 0.f;
    // four k per lane in flight: the loop is bound by the latency of its loads (a row of `a` and
    // N short rows of W from the L2), not by bytes
    int64_t k = lane;
    for (; k + 3 * 64 < K; k += 4 * 64) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = a[r * K + k + 64 * u];
#pragma unroll
      for (int j = 0; j < SKN; ++j) {
        if (j < N) {
          float wv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) wv[u] = w[j * wsj + (k + 64 * u) * wsk];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] = fmaf(v[u], wv[u], acc[j]);
        }
      }
    }
    for (; k < K; k += 64) {
      const float v = a[r * K + k];
#pragma unroll
      for (int j = 0; j < SKN; ++j)
        if (j < N) acc[j] = fmaf(v, w[j * wsj + k * wsk], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < SKN; ++j) {
      if (j < N) {
        const float t = tv_wave_sum(acc[j]);
        if (lane == 0) out[r * N + j] = t + (bias ? bias[j] : 0.f);
      }
    }
  }
}

// thread per large index k: out[r,k] = sum_j d[r,j] W(j,k) (+ bias[k])
__global__ __launch_bounds__(256) void skinny_expand_kernel(int64_t rows, int64_t K, int N,
                                                            int64_t wsj, int64_t wsk,
                                                            const float *__restrict__ w,
                                                            const float *__restrict__ d,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float wk[SKN];
#pragma unroll
  for (int j = 0; j < SKN; ++j) wk[j] = (j < N) ? w[j * wsj + k * wsk] : 0.f;
  const float bk = bias ? bias[k] : 0.f;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    float acc = bk;
#pragma unroll
    for (int j = 0; j < SKN; ++j)
      if (j < N) acc = fmaf(d[r * N + j], wk[j], acc);
    out[r * K + k] = acc;
  }
}

// partial[by][j*osj + k*osk] = sum over this workgroup's rows of d[r,j] a[r,k]; bias partial
// sums behind them at partial[by][N*K + ...]: of d's columns (BIAS_OF_A = false, N entries) or
// of a's columns (true, K entries)
template <bool BIAS_OF_A>
__global__ __launch_bounds__(256) void skinny_outer_kernel(int64_t rows, int64_t K, int N,
                                                           int64_t osj, int64_t osk,
                                                           const float *__restrict__ a,
                                                           const float *__restrict__ d,
                                                           float *__restrict__ partial) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nblk = gridDim.y, by = blockIdx.y;
  const int64_t slot = (int64_t)N * K + (BIAS_OF_A ? K : N);
  if (!BIAS_OF_A && blockIdx.x == 0 && threadIdx.x < N) {
    float b = 0.f;
    for (int64_t r = by; r < rows; r += nblk) b += d[r * N + threadIdx.x];
    partial[by * slot + (int64_t)N * K + threadIdx.x] = b;
  }
  if (k >= K) return;
  float acc[SKN];
#pragma unroll
  for (int j = 0; j < SKN; ++j) acc[j] = 0.f;
  float bsum = 0.f;
  for (int64_t r = by; r < rows; r += nblk) {
    const float v = a[r * K + k];
    bsum += v;
#pragma unroll
    for (int j = 0; j < SKN; ++j)
      if (j < N) acc[j] = fmaf(v, d[r * N + j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < SKN; ++j)
    if (j < N) partial[by * slot + j * osj + k * osk] = acc[j];
  if (BIAS_OF_A) partial[by * slot + (int64_t)N * K + k] = bsum;
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

namespace {
int skinny_check(const char *who, int64_t rows, int64_t big, int64_t small) {
  TV_REQUIRE(rows >= 0 && big >= 1, "%s: bad sizes", who);
  TV_REQUIRE(small >= 1 && small <= SKN, "%s: the small dimension must be in [1,%d]: got %lld", who,
             SKN, (long long)small);
  return TV_OK;
}
unsigned row_wave_blocks(int64_t rows) {
  int64_t blocks = tv_cdiv(rows, 4);
  return (unsigned)(blocks > 8192 ? 8192 : blocks);
}
}  // namespace

extern "C" int tv_skinny_linear_fwd(int64_t rows, int64_t in_features, int64_t out_features,
                                    const float *x, const float *w, const float *bias, float *out,
                                    void *stream) {
  const bool few_out = out_features <= in_features;
  const int64_t n = few_out ? out_features : in_features, k = few_out ? in_features : out_features;
  if (int rc = skinny_check("tv_skinny_linear_fwd", rows, k, n)) return rc;
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && w && out, "tv_skinny_linear_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (few_out) {
    hipLaunchKernelGGL(skinny_rowdot_kernel, dim3(row_wave_blocks(rows)), dim3(256), 0, s, rows, k,
                       (int)n, k, (int64_t)1, x, w, bias, out);
  } else {
    const int64_t nblk = tv_stat_blocks_host(rows);
    hipLaunchKernelGGL(skinny_expand_kernel, dim3((unsigned)tv_cdiv(k, 256), (unsigned)nblk), dim3(256),
                       0, s, rows, k, (int)n, (int64_t)1, n, w, x, bias, out);
  }
  TV_LAUNCH_CHECK("tv_skinny_linear_fwd");
  return TV_OK;
}

extern "C" int64_t tv_skinny_linear_bwd_workspace_bytes(int64_t rows, int64_t in_features,
                                                        int64_t out_features) {
  return (int64_t)sizeof(float) * tv_stat_blocks_host(rows) *
         (in_features * out_features + out_features);
}

extern "C" int tv_skinny_linear_bwd(int64_t rows, int64_t in_features, int64_t out_features,
                                    const float *x, const float *w, const float *dout, float *dx,
                                    float *dwb, void *workspace, int64_t workspace_bytes,
                                    void *stream) {
  const bool few_out = out_features <= in_features;
  const int64_t n = few_out ? out_features : in_features, k = few_out ? in_features : out_features;
  if (int rc = skinny_check("tv_skinny_linear_bwd", rows, k, n)) return rc;
  TV_REQUIRE(rows >= 1, "tv_skinny_linear_bwd: no rows");
  TV_REQUIRE(x && w && dout && dwb, "tv_skinny_linear_bwd: null pointer");
  const int64_t need = tv_skinny_linear_bwd_workspace_bytes(rows, in_features, out_features);
  if (!workspace || workspace_bytes < need) {
    tv_set_error("tv_skinny_linear_bwd: workspace of %lld bytes needed, %lld given",
                 (long long)need, (long long)workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t nblk = tv_stat_blocks_host(rows);
  const dim3 kgrid((unsigned)tv_cdiv(k, 256), (unsigned)nblk);
  const int64_t total = in_features * out_features + out_features;
  if (few_out) {   // W [n, k]: dx[r,k] = sum_j dout[r,j] W(j,k); dW[j,k] = sum_r dout[r,j] x[r,k]
    if (dx) {
      hipLaunchKernelGGL(skinny_expand_kernel, kgrid, dim3(256), 0, s, rows, k, (int)n, k,
                         (int64_t)1, w, dout, (const float *)nullptr, dx);
      TV_LAUNCH_CHECK("tv_skinny_linear_bwd(dx)");
    }
    hipLaunchKernelGGL(skinny_outer_kernel<false>, kgrid, dim3(256), 0, s, rows, k, (int)n, k,
                       (int64_t)1, x, dout, (float *)workspace);
  } else {         // W [k, n]: dx[r,j] = sum_k dout[r,k] W(j,k); dW[k,j] = sum_r dout[r,k] x[r,j]
    if (dx) {
      hipLaunchKernelGGL(skinny_rowdot_kernel, dim3(row_wave_blocks(rows)), dim3(256), 0, s, rows, k,
                         (int)n, (int64_t)1, n, dout, w, (const float *)nullptr, dx);
      TV_LAUNCH_CHECK("tv_skinny_linear_bwd(dx)");
    }
    hipLaunchKernelGGL(skinny_outer_kernel<true>, kgrid, dim3(256), 0, s, rows, k, (int)n,
                       (int64_t)1, n, dout, x, (float *)workspace);
  }
  TV_LAUNCH_CHECK("tv_skinny_linear_bwd(dw)");
  hipLaunchKernelGGL(sum_slots_kernel, dim3(tv_slot_grid(total)), dim3(256), 0, s, nblk, total,
                     (const float *)workspace, dwb);
  TV_LAUNCH_CHECK("tv_skinny_linear_bwd(reduce)");
  return TV_OK;
}
