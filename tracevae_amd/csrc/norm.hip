// BatchNorm over rows + LeakyReLU (training: batch statistics, eval: folded running
// statistics), forward and backward.  HBM-bound streaming kernels: a thread owns a column
// (column reductions, rows strided over a fixed number of workgroups) or a float4 of a row
// (elementwise passes).  Column statistics are reduced in two fixed-order stages, the
// second one in float64.
// Algorithmic bytes per row of C floats: stats 4C (read), apply 8C (read+write),
// backward 8C (reduce: x, dy) + 12C (apply: x, dy, dx).
#include "tv_common.h"
#include "tv_slots.h"

namespace {

__global__ __launch_bounds__(256) void colstats_kernel(int64_t rows, int64_t C,
                                                       const float *__restrict__ x,
                                                       const float *__restrict__ row_weight,
                                                       float *__restrict__ partial) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nblk = gridDim.y, by = blockIdx.y;
  if (col >= C) return;
  float s = 0.f, q = 0.f;
  for (int64_t r = by; r < rows; r += nblk) {
    const float v = x[r * C + col];
    const float wv = row_weight ? row_weight[r] * v : v;
    s += wv;
    q = fmaf(wv, v, q);
  }
  partial[(by * 2 + 0) * C + col] = s;
  partial[(by * 2 + 1) * C + col] = q;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(int64_t C, int64_t nblk,
                                                          const float *__restrict__ partial,
                                                          double count, float eps, float momentum,
                                                          float *mean, float *rstd,
                                                          float *running_mean,
                                                          float *running_var,
                                                          const float *gamma, const float *beta,
                                                          float *scale, float *shift,
                                                          float *act_max) {
  const int64_t col = (int64_t)blockIdx.x * TV_SLOT_COLS + threadIdx.x % TV_SLOT_COLS;
  double tot[2];
  tv_slot_sums<2>(partial, nblk, 2 * C, C, col, col < C, tot);
  if (act_max) {
    // a bound on |gamma * xhat + beta| over the rows that entered the statistics, for the fp16
    // GEMM forms that apply this BatchNorm + LeakyReLU to their operand (tv_gemm_f32_ex, state
    // 2): sum_r w_r (x_r - mean)^2 = count * var and every weight is >= 1, so |xhat| <=
    // sqrt(count); |lrelu(y)| <= |y|.  Loose (sqrt(count) against the ~5 sigma of real data: 5-6
    // binades of the 29 the fp16 split has to spare), but it costs no pass over the rows.  One
    // slot per workgroup (at most 256 of them: C <= 2048), the rest zero.
    float b = 0.f;
    if (threadIdx.x < TV_SLOT_COLS && col < C) b = fabsf(gamma[col]) * (float)sqrt(count) + fabsf(beta[col]);
    if (threadIdx.x < 64) {
      b = tv_wave_max(b);
      if (threadIdx.x == 0) act_max[blockIdx.x] = b;
    }
    if (blockIdx.x == 0)
      for (int sl = (int)gridDim.x + threadIdx.x; sl < 256; sl += 256) act_max[sl] = 0.f;
  }
  if (threadIdx.x >= TV_SLOT_COLS || col >= C) return;
  const double m = tot[0] / count;
  double var = tot[1] / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[col] = (float)m;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  rstd[col] = rs;
  if (scale) {          // y = x*scale + shift, the folded form the GEMM prologue applies
    const float sc = gamma[col] * rs;
    scale[col] = sc;
    shift[col] = beta[col] - (float)m * sc;
  }
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[col] = (float)((1.0 - momentum) * (double)running_mean[col] + momentum * m);
    running_var[col] = (float)((1.0 - momentum) * (double)running_var[col] + momentum * unbiased);
  }
}

__global__ __launch_bounds__(256) void bn_eval_coeffs_kernel(int64_t C, const float *rm,
                                                             const float *rv, const float *gamma,
                                                             const float *beta, float eps,
                                                             float *scale, float *shift) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= C) return;
  const float sc = gamma[col] / sqrtf(rv[col] + eps);
  scale[col] = sc;
  shift[col] = beta[col] - rm[col] * sc;
}

// y = lrelu(x * sc[col] + sh[col]); TRAIN: sc = rstd*gamma, sh = beta - mean*sc computed here
template <bool VEC, bool TRAIN>
__global__ __launch_bounds__(256) void affine_lrelu_kernel(int64_t rows, int64_t C,
                                                           const float *__restrict__ x,
                                                           const float *__restrict__ p0,
                                                           const float *__restrict__ p1,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta,
                                                           float slope, float *__restrict__ y) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC) {
    const int64_t c4 = C >> 2, total = rows * c4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
      const int64_t q = e % c4;
      const float4 v = reinterpret_cast<const float4 *>(x)[e];
      float4 a = reinterpret_cast<const float4 *>(p0)[q];
      float4 b = reinterpret_cast<const float4 *>(p1)[q];
      if (TRAIN) {  // p0 = mean, p1 = rstd
        const float4 g = reinterpret_cast<const float4 *>(gamma)[q];
        const float4 bt = reinterpret_cast<const float4 *>(beta)[q];
        const float4 sc = make_float4(b.x * g.x, b.y * g.y, b.z * g.z, b.w * g.w);
        b = make_float4(bt.x - a.x * sc.x, bt.y - a.y * sc.y, bt.z - a.z * sc.z, bt.w - a.w * sc.w);
        a = sc;
      }
      float4 o;
      o.x = tv_lrelu(fmaf(v.x, a.x, b.x), slope);
      o.y = tv_lrelu(fmaf(v.y, a.y, b.y), slope);
      o.z = tv_lrelu(fmaf(v.z, a.z, b.z), slope);
      o.w = tv_lrelu(fmaf(v.w, a.w, b.w), slope);
      reinterpret_cast<float4 *>(y)[e] = o;
    }
  } else {
    const int64_t total = rows * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
      const int64_t c = e % C;
      float a = p0[c], b = p1[c];
      if (TRAIN) {
        const float sc = b * gamma[c];
        b = beta[c] - a * sc;
        a = sc;
      }
      y[e] = tv_lrelu(fmaf(x[e], a, b), slope);
    }
  }
}

// backward, stage 1: partial column sums of dyl = dy*lrelu'(pre) and dyl*xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    int64_t rows, int64_t C, const float *__restrict__ x, const float *__restrict__ dy,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
    float *__restrict__ partial) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nblk = gridDim.y, by = blockIdx.y;
  if (col >= C) return;
  const float m = mean[col], rs = rstd[col], g = gamma[col], b = beta[col];
  float s1 = 0.f, s2 = 0.f;
  for (int64_t r = by; r < rows; r += nblk) {
    const float xh = (x[r * C + col] - m) * rs;
    const float pre = fmaf(g, xh, b);
    const float dyl = dy[r * C + col] * (pre > 0.f ? 1.f : slope);
    s1 += dyl;
    s2 = fmaf(dyl, xh, s2);
  }
  partial[(by * 2 + 0) * C + col] = s1;
  partial[(by * 2 + 1) * C + col] = s2;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(int64_t C, int64_t nblk,
                                                              const float *__restrict__ partial,
                                                              float *sums, float *dgamma,
                                                              float *dbeta) {
  const int64_t col = (int64_t)blockIdx.x * TV_SLOT_COLS + threadIdx.x % TV_SLOT_COLS;
  double tot[2];
  tv_slot_sums<2>(partial, nblk, 2 * C, C, col, col < C, tot);
  if (threadIdx.x >= TV_SLOT_COLS || col >= C) return;
  sums[col] = (float)tot[0];
  sums[C + col] = (float)tot[1];
  dbeta[col] = (float)tot[0];
  dgamma[col] = (float)tot[1];
}

// backward, stage 2: dx = gamma*rstd*(dyl - mean(dyl) - xhat*mean(dyl*xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    int64_t rows, int64_t C, const float *__restrict__ x, const float *__restrict__ dy,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
    const float *__restrict__ sums, const float *__restrict__ row_weight, float inv_n,
    float *__restrict__ dx) {
  const int64_t total = rows * C, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t c = e % C;
    const float rs = rstd[c], g = gamma[c];
    const float xh = (x[e] - mean[c]) * rs;
    const float pre = fmaf(g, xh, beta[c]);
    const float dyl = dy[e] * (pre > 0.f ? 1.f : slope);
    // a row of weight w stands for w identical rows: the statistics move w times as much
    const float wn = row_weight ? row_weight[e / C] * inv_n : inv_n;
    dx[e] = g * rs * (dyl - sums[c] * wn - xh * sums[C + c] * wn);
  }
}

// the same two stages on float4 columns (C % 4 == 0, 16-byte aligned operands): a thread owns
// four columns, keeps their coefficients in registers and walks rows with 16-byte accesses.
// Row order and arithmetic per element are those of the scalar kernels (bit-identical results).
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(
    int64_t rows, int64_t C, const float *__restrict__ x, const float *__restrict__ dy,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
    float *__restrict__ partial) {
  const int64_t c4 = C >> 2;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nblk = gridDim.y, by = blockIdx.y;
  if (q >= c4) return;
  const float4 m = reinterpret_cast<const float4 *>(mean)[q];
  const float4 rs = reinterpret_cast<const float4 *>(rstd)[q];
  const float4 g = reinterpret_cast<const float4 *>(gamma)[q];
  const float4 b = reinterpret_cast<const float4 *>(beta)[q];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const float mm[4] = {m.x, m.y, m.z, m.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w};
  const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll 4
  for (int64_t r = by; r < rows; r += nblk) {
    const float4 xv = reinterpret_cast<const float4 *>(x)[r * c4 + q];
    const float4 dv = reinterpret_cast<const float4 *>(dy)[r * c4 + q];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xs[j] - mm[j]) * rr[j];
      const float pre = fmaf(gg[j], xh, bb[j]);
      const float dyl = ds[j] * (pre > 0.f ? 1.f : slope);
      s1[j] += dyl;
      s2[j] = fmaf(dyl, xh, s2[j]);
    }
  }
  reinterpret_cast<float4 *>(partial + (by * 2 + 0) * C)[q] = make_float4(s1[0], s1[1], s1[2], s1[3]);
  reinterpret_cast<float4 *>(partial + (by * 2 + 1) * C)[q] = make_float4(s2[0], s2[1], s2[2], s2[3]);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(
    int64_t rows, int64_t C, const float *__restrict__ x, const float *__restrict__ dy,
    const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
    const float *__restrict__ sums, const float *__restrict__ row_weight, float inv_n,
    float *__restrict__ dx) {
  const int64_t c4 = C >> 2;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= c4) return;
  const float4 m = reinterpret_cast<const float4 *>(mean)[q];
  const float4 rs = reinterpret_cast<const float4 *>(rstd)[q];
  const float4 g = reinterpret_cast<const float4 *>(gamma)[q];
  const float4 b = reinterpret_cast<const float4 *>(beta)[q];
  const float4 t1 = reinterpret_cast<const float4 *>(sums)[q];
  const float4 t2 = reinterpret_cast<const float4 *>(sums + C)[q];
  const float mm[4] = {m.x, m.y, m.z, m.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w};
  const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
  const float u1[4] = {t1.x, t1.y, t1.z, t1.w}, u2[4] = {t2.x, t2.y, t2.z, t2.w};
#pragma unroll 4
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float4 xv = reinterpret_cast<const float4 *>(x)[r * c4 + q];
    const float4 dv = reinterpret_cast<const float4 *>(dy)[r * c4 + q];
    const float wn = row_weight ? row_weight[r] * inv_n : inv_n;
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xs[j] - mm[j]) * rr[j];
      const float pre = fmaf(gg[j], xh, bb[j]);
      const float dyl = ds[j] * (pre > 0.f ? 1.f : slope);
      o[j] = gg[j] * rr[j] * (dyl - u1[j] * wn - xh * u2[j] * wn);
    }
    reinterpret_cast<float4 *>(dx)[r * c4 + q] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

__global__ __launch_bounds__(256) void affine_lrelu_bwd_kernel(
    int64_t rows, int64_t C, const float *__restrict__ x, const float *__restrict__ dy,
    const float *__restrict__ scale, const float *__restrict__ shift, float slope,
    float *__restrict__ dx) {
  const int64_t total = rows * C, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t c = e % C;
    const float pre = fmaf(x[e], scale[c], shift[c]);
    dx[e] = dy[e] * (pre > 0.f ? 1.f : slope) * scale[c];
  }
}

inline unsigned ew_blocks(int64_t work_items) {
  int64_t b = tv_cdiv(work_items, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline bool vec_ok(int64_t C, const void *a, const void *b) {
  return (C % 4 == 0) && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}

}  // namespace

extern "C" int tv_colstats(int64_t rows, int64_t cols, const float *x, const float *row_weight,
                           float *partial, void *stream) {
  TV_REQUIRE(rows >= 1 && cols >= 1, "tv_colstats: empty input");
  TV_REQUIRE(x && partial, "tv_colstats: null pointer");
  const dim3 grid((unsigned)tv_cdiv(cols, 256), (unsigned)tv_stat_blocks_host(rows));
  hipLaunchKernelGGL(colstats_kernel, grid, dim3(256), 0, (hipStream_t)stream, rows, cols, x,
                     row_weight, partial);
  TV_LAUNCH_CHECK("tv_colstats");
  return TV_OK;
}

extern "C" int tv_bn_finalize(int64_t cols, int64_t nblk, const float *partial, int64_t count,
                              float eps, float momentum, float *mean, float *rstd,
                              float *running_mean, float *running_var, void *stream) {
  TV_REQUIRE(cols >= 1 && nblk >= 1 && count >= 1, "tv_bn_finalize: bad sizes");
  TV_REQUIRE(partial && mean && rstd, "tv_bn_finalize: null pointer");
  TV_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
             "tv_bn_finalize: running_mean and running_var go together");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(tv_slot_grid(cols)), dim3(256), 0,
                     (hipStream_t)stream, cols, nblk, partial, (double)count, eps, momentum, mean,
                     rstd, running_mean, running_var, (const float *)nullptr,
                     (const float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr);
  TV_LAUNCH_CHECK("tv_bn_finalize");
  return TV_OK;
}

extern "C" int tv_bn_finalize_coeffs(int64_t cols, int64_t nblk, const float *partial,
                                     int64_t count, float eps, float momentum, float *mean,
                                     float *rstd, float *running_mean, float *running_var,
                                     const float *gamma, const float *beta, float *scale,
                                     float *shift, float *act_max, void *stream) {
  TV_REQUIRE(cols >= 1 && nblk >= 1 && count >= 1, "tv_bn_finalize_coeffs: bad sizes");
  TV_REQUIRE(!act_max || tv_slot_grid(cols) <= 256,
             "tv_bn_finalize_coeffs: act_max takes at most 2048 columns");
  TV_REQUIRE(partial && mean && rstd && gamma && beta && scale && shift,
             "tv_bn_finalize_coeffs: null pointer");
  TV_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
             "tv_bn_finalize_coeffs: running_mean and running_var go together");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(tv_slot_grid(cols)), dim3(256), 0,
                     (hipStream_t)stream, cols, nblk, partial, (double)count, eps, momentum, mean,
                     rstd, running_mean, running_var, gamma, beta, scale, shift, act_max);
  TV_LAUNCH_CHECK("tv_bn_finalize_coeffs");
  return TV_OK;
}

extern "C" int tv_bn_apply_lrelu(int64_t rows, int64_t cols, const float *x, const float *mean,
                                 const float *rstd, const float *gamma, const float *beta,
                                 float slope, float *y, void *stream) {
  TV_REQUIRE(rows >= 0 && cols >= 1, "tv_bn_apply_lrelu: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && mean && rstd && gamma && beta && y, "tv_bn_apply_lrelu: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (vec_ok(cols, x, y) && vec_ok(cols, mean, rstd) && vec_ok(cols, gamma, beta))
    hipLaunchKernelGGL((affine_lrelu_kernel<true, true>), dim3(ew_blocks(rows * cols / 4)),
                       dim3(256), 0, s, rows, cols, x, mean, rstd, gamma, beta, slope, y);
  else
    hipLaunchKernelGGL((affine_lrelu_kernel<false, true>), dim3(ew_blocks(rows * cols)), dim3(256),
                       0, s, rows, cols, x, mean, rstd, gamma, beta, slope, y);
  TV_LAUNCH_CHECK("tv_bn_apply_lrelu");
  return TV_OK;
}

extern "C" int tv_bn_eval_coeffs(int64_t cols, const float *running_mean, const float *running_var,
                                 const float *gamma, const float *beta, float eps, float *scale,
                                 float *shift, void *stream) {
  TV_REQUIRE(cols >= 1, "tv_bn_eval_coeffs: bad size");
  TV_REQUIRE(running_mean && running_var && gamma && beta && scale && shift,
             "tv_bn_eval_coeffs: null pointer");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((unsigned)tv_cdiv(cols, 256)), dim3(256), 0,
                     (hipStream_t)stream, cols, running_mean, running_var, gamma, beta, eps, scale,
                     shift);
  TV_LAUNCH_CHECK("tv_bn_eval_coeffs");
  return TV_OK;
}

extern "C" int tv_affine_lrelu(int64_t rows, int64_t cols, const float *x, const float *scale,
                               const float *shift, float slope, float *y, void *stream) {
  TV_REQUIRE(rows >= 0 && cols >= 1, "tv_affine_lrelu: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && scale && shift && y, "tv_affine_lrelu: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (vec_ok(cols, x, y) && vec_ok(cols, scale, shift))
    hipLaunchKernelGGL((affine_lrelu_kernel<true, false>), dim3(ew_blocks(rows * cols / 4)),
                       dim3(256), 0, s, rows, cols, x, scale, shift, nullptr, nullptr, slope, y);
  else
    hipLaunchKernelGGL((affine_lrelu_kernel<false, false>), dim3(ew_blocks(rows * cols)),
                       dim3(256), 0, s, rows, cols, x, scale, shift, nullptr, nullptr, slope, y);
  TV_LAUNCH_CHECK("tv_affine_lrelu");
  return TV_OK;
}

extern "C" int64_t tv_bn_lrelu_bwd_workspace_bytes(int64_t rows, int64_t cols) {
  return (int64_t)sizeof(float) * (tv_stat_blocks_host(rows) * 2 * cols + 2 * cols);
}

extern "C" int tv_bn_lrelu_bwd(int64_t rows, int64_t cols, const float *x, const float *dy,
                               const float *mean, const float *rstd, const float *gamma,
                               const float *beta, float slope, const float *row_weight,
                               int64_t count, float *dx, float *dgamma, float *dbeta,
                               void *workspace, int64_t workspace_bytes, void *stream) {
  TV_REQUIRE(rows >= 1 && cols >= 1 && count >= 1, "tv_bn_lrelu_bwd: bad sizes");
  TV_REQUIRE(row_weight || count == rows, "tv_bn_lrelu_bwd: count must equal rows without weights");
  TV_REQUIRE(x && dy && mean && rstd && gamma && beta && dx && dgamma && dbeta,
             "tv_bn_lrelu_bwd: null pointer");
  const int64_t need = tv_bn_lrelu_bwd_workspace_bytes(rows, cols);
  if (!workspace || workspace_bytes < need) {
    tv_set_error("tv_bn_lrelu_bwd: workspace of %lld bytes needed, %lld given", (long long)need,
                 (long long)workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t nblk = tv_stat_blocks_host(rows);
  float *partial = (float *)workspace;
  float *sums = partial + nblk * 2 * cols;
  // `sums` sits behind nblk*2*cols floats of a 16-byte aligned workspace: aligned when cols % 4 == 0
  const bool vec = vec_ok(cols, x, dy) && vec_ok(cols, mean, rstd) && vec_ok(cols, gamma, beta) &&
                   vec_ok(cols, dx, workspace);
  if (vec)
    hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel, dim3((unsigned)tv_cdiv(cols / 4, 256), (unsigned)nblk),
                       dim3(256), 0, s, rows, cols, x, dy, mean, rstd, gamma, beta, slope, partial);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)tv_cdiv(cols, 256), (unsigned)nblk),
                       dim3(256), 0, s, rows, cols, x, dy, mean, rstd, gamma, beta, slope, partial);
  TV_LAUNCH_CHECK("tv_bn_lrelu_bwd(reduce)");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(tv_slot_grid(cols)), dim3(256), 0, s,
                     cols, nblk, partial, sums, dgamma, dbeta);
  TV_LAUNCH_CHECK("tv_bn_lrelu_bwd(finalize)");
  if (vec) {
    const int64_t cblocks = tv_cdiv(cols / 4, 256);
    int64_t rblocks = tv_cdiv(4096, cblocks);
    if (rblocks > rows) rblocks = rows;
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3((unsigned)cblocks, (unsigned)rblocks), dim3(256),
                       0, s, rows, cols, x, dy, mean, rstd, gamma, beta, slope, sums, row_weight,
                       (float)(1.0 / (double)count), dx);
  } else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, s, rows,
                       cols, x, dy, mean, rstd, gamma, beta, slope, sums, row_weight,
                       (float)(1.0 / (double)count), dx);
  TV_LAUNCH_CHECK("tv_bn_lrelu_bwd(apply)");
  return TV_OK;
}

extern "C" int tv_affine_lrelu_bwd(int64_t rows, int64_t cols, const float *x, const float *dy,
                                   const float *scale, const float *shift, float slope, float *dx,
                                   void *stream) {
  TV_REQUIRE(rows >= 0 && cols >= 1, "tv_affine_lrelu_bwd: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && dy && scale && shift && dx, "tv_affine_lrelu_bwd: null pointer");
  hipLaunchKernelGGL(affine_lrelu_bwd_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0,
                     (hipStream_t)stream, rows, cols, x, dy, scale, shift, slope, dx);
  TV_LAUNCH_CHECK("tv_affine_lrelu_bwd");
  return TV_OK;
}
