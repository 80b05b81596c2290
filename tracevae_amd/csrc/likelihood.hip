// (c)/(d) likelihood kernels: adjacency-logit contraction fused with the Bernoulli
// log-likelihood, categorical / normal log-likelihoods with the reference's Biased / Safe /
// Masked variants, reparameterised sampling, and the K-sample score reduction.
// All sums are fixed-order (per-thread partials -> wave / workgroup tree), no atomics.
#include "tv_common.h"

namespace {

constexpr float LOG_2PI = 1.8378770664093453f;

// ------------------------------------------------------------------------------------
// p(adj | z): l_ij = <h_i, h_j> for i < j, Bernoulli log-likelihood over all T pairs.
// One workgroup per (sample, trace).  64x64 pair tiles, 4x4 pairs per thread; the two
// 64-row slices of h sit in LDS as [d][row] (row stride 68) so the operand reads are
// float4 and conflict-free.  [K,B,M,M] is never materialised.
// Algorithmic bytes per (k,b): 4*M*D (h) + 4*T (adj, L2-resident across the K samples).
// ------------------------------------------------------------------------------------
constexpr int PT = 64;         // pair-tile edge
constexpr int HS = 68;         // LDS row stride of the h slices

__device__ __forceinline__ void load_h_slice(float *dst, const float *h, int64_t row0, int64_t M,
                                             int D) {
  // dst[d][r] = h[row0 + r][d], zero beyond M
  for (int e = threadIdx.x; e < PT * D; e += blockDim.x) {
    const int r = e / D, d = e - r * D;
    dst[d * HS + r] = (row0 + r < M) ? h[(row0 + r) * D + d] : 0.f;
  }
}

__device__ __forceinline__ void pair_logits(float (&l)[4][4], const float *hi, const float *hj,
                                            int D, int ty, int tx) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) l[a][b] = 0.f;
  for (int d = 0; d < D; ++d) {
    const float4 x = *reinterpret_cast<const float4 *>(&hi[d * HS + ty * 4]);
    const float4 y = *reinterpret_cast<const float4 *>(&hj[d * HS + tx * 4]);
    const float xa[4] = {x.x, x.y, x.z, x.w}, yb[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) l[a][b] = fmaf(xa[a], yb[b], l[a][b]);
  }
}

__device__ __forceinline__ int64_t triu_index(int64_t i, int64_t j, int64_t M) {
  return i * (2 * M - i - 1) / 2 + (j - i - 1);
}

__global__ __launch_bounds__(256) void adj_ll_fwd_kernel(int64_t B, int64_t M, int D,
                                                         const float *__restrict__ h,
                                                         const float *__restrict__ adj,
                                                         int biased, float alpha, float log_thr,
                                                         float *__restrict__ ll) {
  extern __shared__ __align__(16) float lds[];
  float *hi = lds, *hj = lds + D * HS;
  __shared__ float red[256];
  const int64_t kb = blockIdx.x, b = kb % B;
  const float *hb = h + kb * M * D;
  const float *ab = adj + b * (M * (M - 1) / 2);
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int nt = (int)((M + PT - 1) / PT);
  float acc = 0.f;
  for (int ti = 0; ti < nt; ++ti) {
    __syncthreads();
    load_h_slice(hi, hb, (int64_t)ti * PT, M, D);
    for (int tj = ti; tj < nt; ++tj) {
      __syncthreads();
      load_h_slice(hj, hb, (int64_t)tj * PT, M, D);
      __syncthreads();
      float l[4][4];
      pair_logits(l, hi, hj, D, ty, tx);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t gi = (int64_t)ti * PT + ty * 4 + a, gj = (int64_t)tj * PT + tx * 4 + c;
          if (gi < gj && gj < M) {
            const float x = ab[triu_index(gi, gj, M)];
            float lp = x * l[a][c] - tv_softplus(l[a][c]);
            if (biased && lp < log_thr) lp *= alpha;
            acc += lp;
          }
        }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ll[kb] = red[0];
}

// dh_i = sum_{j != i} g_ij h_j,  g_ij = dll * f'(lp_ij) * (adj_ij - sigmoid(l_ij))
template <int DCH>  // DCH >= ceil(D/4)
__global__ __launch_bounds__(256) void adj_ll_bwd_kernel(int64_t B, int64_t M, int D,
                                                         const float *__restrict__ h,
                                                         const float *__restrict__ adj,
                                                         int biased, float alpha, float log_thr,
                                                         const float *__restrict__ dll,
                                                         float *__restrict__ dh) {
  extern __shared__ __align__(16) float lds[];
  float *hi = lds, *hj = lds + D * HS, *gs = lds + 2 * D * HS;  // gs[64][65]
  const int64_t kb = blockIdx.x, b = kb % B;
  const float *hb = h + kb * M * D;
  const float *ab = adj + b * (M * (M - 1) / 2);
  const float dl = dll[kb];
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int ri = threadIdx.x >> 2, part = threadIdx.x & 3;  // phase 2: row, d = part + 4*dd
  const int nt = (int)((M + PT - 1) / PT);
  for (int ti = 0; ti < nt; ++ti) {
    float acc[DCH];
#pragma unroll
    for (int dd = 0; dd < DCH; ++dd) acc[dd] = 0.f;
    __syncthreads();
    load_h_slice(hi, hb, (int64_t)ti * PT, M, D);
    for (int tj = 0; tj < nt; ++tj) {
      __syncthreads();
      load_h_slice(hj, hb, (int64_t)tj * PT, M, D);
      __syncthreads();
      float l[4][4];
      pair_logits(l, hi, hj, D, ty, tx);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t gi = (int64_t)ti * PT + ty * 4 + a, gj = (int64_t)tj * PT + tx * 4 + c;
          float g = 0.f;
          if (gi != gj && gi < M && gj < M) {
            const int64_t lo = gi < gj ? gi : gj, hi_ = gi < gj ? gj : gi;
            const float x = ab[triu_index(lo, hi_, M)];
            const float lv = l[a][c];
            const float lp = x * lv - tv_softplus(lv);
            const float sg = 1.0f / (1.0f + expf(-lv));
            g = dl * (x - sg);
            if (biased && lp < log_thr) g *= alpha;
          }
          gs[(ty * 4 + a) * 65 + tx * 4 + c] = g;
        }
      __syncthreads();
#pragma unroll
      for (int dd = 0; dd < DCH; ++dd) {
        const int d = part + 4 * dd;
        if (d < D) {
          float s = acc[dd];
          for (int j = 0; j < PT; ++j) s = fmaf(gs[ri * 65 + j], hj[d * HS + j], s);
          acc[dd] = s;
        }
      }
    }
    const int64_t gi = (int64_t)ti * PT + ri;
    if (gi < M) {
#pragma unroll
      for (int dd = 0; dd < DCH; ++dd) {
        const int d = part + 4 * dd;
        if (d < D) dh[(kb * M + gi) * D + d] = acc[dd];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Categorical: one wave per row, single pass online log-sum-exp
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void categorical_ll_fwd_kernel(
    int64_t rows, int64_t V, int64_t target_rows, const float *__restrict__ logits,
    const int64_t *__restrict__ target, int biased, float alpha, float log_thr,
    float *__restrict__ ll, float *__restrict__ lse_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float *x = logits + r * V;
    float m = -INFINITY, s = 0.f;
    for (int64_t c = lane; c < V; c += 64) {
      const float v = x[c];
      if (v > m) { s = s * expf(m - v) + 1.0f; m = v; } else { s += expf(v - m); }
    }
    const float mm = tv_wave_max(m);
    s = tv_wave_sum(m == -INFINITY ? 0.f : s * expf(m - mm));
    const float lse = mm + logf(s);
    if (lane == 0) {
      int64_t t = target[r % target_rows];
      t = t < 0 ? 0 : (t >= V ? V - 1 : t);
      float lp = x[t] - lse;
      if (biased && lp < log_thr) lp *= alpha;
      ll[r] = lp;
      if (lse_out) lse_out[r] = lse;
    }
  }
}

__global__ __launch_bounds__(256) void categorical_ll_bwd_kernel(
    int64_t rows, int64_t V, int64_t target_rows, const float *__restrict__ logits,
    const int64_t *__restrict__ target, int biased, float alpha, float log_thr,
    const float *__restrict__ lse, const float *__restrict__ dll, float *__restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float *x = logits + r * V;
    float *dx = dlogits + r * V;
    int64_t t = target[r % target_rows];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const float ls = lse[r];
    float g = dll[r];
    if (biased && (x[t] - ls) < log_thr) g *= alpha;
    for (int64_t c = lane; c < V; c += 64)
      dx[c] = g * ((c == t ? 1.0f : 0.0f) - expf(x[c] - ls));
  }
}

// ------------------------------------------------------------------------------------
// Normal: reparameterised sample + log q, plain log-likelihood
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float clampf(float x, float lo, float hi) {
  return fminf(fmaxf(x, lo), hi);
}

__global__ __launch_bounds__(256) void normal_reparam_fwd_kernel(
    int64_t K, int64_t B, int64_t D, const float *__restrict__ mean,
    const float *__restrict__ ls_raw, float lo, float hi, const float *__restrict__ eps,
    float *__restrict__ z, float *__restrict__ logq, float *__restrict__ ls_out) {
  const int64_t kb = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (kb >= K * B) return;
  const int64_t b = kb % B;
  float acc = 0.f;
  for (int64_t d = 0; d < D; ++d) {
    const float mu = mean[b * D + d];
    const float ls = clampf(ls_raw[b * D + d], lo, hi);
    const float zz = mu + expf(ls) * eps[kb * D + d];
    z[kb * D + d] = zz;
    const float diff = zz - mu;
    acc += -0.5f * LOG_2PI - ls - 0.5f * diff * diff * expf(-2.0f * ls);
    if (kb < B) ls_out[b * D + d] = ls;
  }
  logq[kb] = acc;
}

__global__ __launch_bounds__(256) void normal_reparam_bwd_kernel(
    int64_t K, int64_t B, int64_t D, const float *__restrict__ ls_raw, float lo, float hi,
    const float *__restrict__ eps, const float *__restrict__ dz, const float *__restrict__ dlogq,
    float *__restrict__ dmean, float *__restrict__ dls_raw) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * D) return;
  const int64_t b = e / D, d = e - b * D;
  const float raw = ls_raw[e];
  const float ls = clampf(raw, lo, hi);
  const float sd = expf(ls);
  float gm = 0.f, gl = 0.f;
  for (int64_t k = 0; k < K; ++k) {
    const int64_t kb = k * B + b;
    const float g = dz ? dz[kb * D + d] : 0.f;
    gm += g;
    // d logq / d logstd = -1 exactly once the reparameterised z is substituted
    gl += g * sd * eps[kb * D + d] - (dlogq ? dlogq[kb] : 0.f);
  }
  dmean[e] = gm;
  dls_raw[e] = (raw >= lo && raw <= hi) ? gl : 0.f;
}

__global__ __launch_bounds__(256) void normal_ll_fwd_kernel(int64_t rows, int64_t D,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ ls_raw,
                                                            float lo, float hi,
                                                            float *__restrict__ ll) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float acc = 0.f;
  for (int64_t d = 0; d < D; ++d) {
    const float ls = clampf(ls_raw[r * D + d], lo, hi);
    const float diff = x[r * D + d] - mean[r * D + d];
    acc += -0.5f * LOG_2PI - ls - 0.5f * diff * diff * expf(-2.0f * ls);
  }
  ll[r] = acc;
}

__global__ __launch_bounds__(256) void normal_ll_bwd_kernel(
    int64_t rows, int64_t D, const float *__restrict__ x, const float *__restrict__ mean,
    const float *__restrict__ ls_raw, float lo, float hi, const float *__restrict__ dll,
    float *__restrict__ dx, float *__restrict__ dmean, float *__restrict__ dls_raw) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * D) return;
  const int64_t r = e / D;
  const float raw = ls_raw[e];
  const float ls = clampf(raw, lo, hi);
  const float diff = x[e] - mean[e];
  const float prec = expf(-2.0f * ls);
  const float g = dll[r];
  if (dx) dx[e] = -g * diff * prec;
  if (dmean) dmean[e] = g * diff * prec;
  if (dls_raw) dls_raw[e] = (raw >= lo && raw <= hi) ? g * (-1.0f + diff * diff * prec) : 0.f;
}

// ------------------------------------------------------------------------------------
// p(latency | z2, G): one wave per (sample, trace), lanes over the M node slots
// ------------------------------------------------------------------------------------
struct LatArgs {
  int64_t K, B, M;
  const float *feats, *latency_pad;
  const int64_t *node_count, *node_type_pad;
  const int64_t *node_offsets;   // non-null: feats rows are compact ([real nodes | one per trace])
  const float *std_limit;
  int has_min;
  float ls_min;
  int mode;
  float thr, alpha, log_bias_thr;
  int use_mask, weighted;
};

template <bool BWD>
__global__ __launch_bounds__(256) void latency_ll_kernel(LatArgs a, float *__restrict__ ll,
                                                         float *__restrict__ std_out,
                                                         const float *__restrict__ dll,
                                                         float *__restrict__ dfeats) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t kb = wave; kb < a.K * a.B; kb += nwaves) {
    const int64_t b = kb % a.B;
    const int64_t n = a.node_count[b];
    const float w = (a.use_mask && a.weighted) ? (float)a.M / (float)n : 1.0f;
    const float g0 = BWD ? dll[kb] * w : 0.f;
    float acc = 0.f;
    // compact rows: replica k holds R = N + B rows, the real nodes of trace b at
    // node_offsets[b].. and the representative of its padded slots at N + b
    int64_t cbase = 0, crep = 0;
    if (a.node_offsets) {
      const int64_t N = a.node_offsets[a.B], k = kb / a.B;
      cbase = k * (N + a.B) + a.node_offsets[b];
      crep = k * (N + a.B) + N + b;
      if (BWD && lane == 0) { dfeats[2 * crep] = 0.f; dfeats[2 * crep + 1] = 0.f; }
    }
    for (int64_t m = lane; m < a.M; m += 64) {
      const int64_t r = kb * a.M + m;
      const int64_t fr = a.node_offsets ? (m < n ? cbase + m : crep) : r;
      const float mu = a.feats[2 * fr], raw = a.feats[2 * fr + 1];
      float ls = raw;
      bool pass = true;  // does d(ls)/d(raw) = 1 ?
      if (a.has_min && ls < a.ls_min) { ls = a.ls_min; pass = false; }
      if (a.std_limit) {
        const float lim = logf(fmaxf(a.std_limit[a.node_type_pad[b * a.M + m]], 1e-7f));
        if (ls > lim) { ls = lim; pass = false; }
      }
      const float sd = expf(ls);
      if (!BWD && std_out) std_out[r] = sd;
      const bool live = !a.use_mask || m < n;
      const float x = a.latency_pad[b * a.M + m];
      float xe = x;
      if (a.mode == 1) xe = fmaxf(fminf(x, mu + a.thr * sd), mu - a.thr * sd);
      const float diff = xe - mu;
      const float prec = expf(-2.0f * ls);
      float lp, scale = 1.0f;
      if (a.mode == 3) {
        const float t = fabsf((x - mu) / sd) - a.thr;
        lp = -0.6931471805599453f + log1pf(-erff(t * 0.7071067811865475f));
        if (lp < a.log_bias_thr) lp *= a.alpha;
      } else {
        lp = -0.5f * LOG_2PI - ls - 0.5f * diff * diff * prec;
        if (a.mode == 2 && fabsf(x - mu) > a.thr * sd) { lp *= a.alpha; scale = a.alpha; }
      }
      if (!BWD) {
        // MaskedDistribution multiplies: log_prob * mask * weight (distributions.py:74-78),
        // so a non-finite log_prob of a masked slot still poisons the sum like it does there
        acc += lp * (live ? 1.0f : 0.0f) * w;
      } else {
        const float g = live ? g0 * scale : 0.f;
        if (!a.node_offsets || m < n) {
          dfeats[2 * fr] = g * diff * prec;
          dfeats[2 * fr + 1] = pass ? g * (-1.0f + diff * diff * prec) : 0.f;
        }
      }
    }
    if (!BWD) {
      acc = tv_wave_sum(acc);
      if (lane == 0) ll[kb] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------
// test-time score reduction over the K samples
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_reduce_kernel(int64_t K, int64_t B,
                                                           const float *__restrict__ w,
                                                           int has_clip, float clip,
                                                           float *__restrict__ loss,
                                                           float *__restrict__ nll) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float mx = -INFINITY, sum = 0.f;
  for (int64_t k = 0; k < K; ++k) {
    const float v = w[k * B + b];
    mx = fmaxf(mx, v);
    sum += v;
  }
  float se = 0.f;
  for (int64_t k = 0; k < K; ++k) se += expf(w[k * B + b] - mx);
  float l = -(sum / (float)K);
  float n = -((mx + logf(se)) - logf((float)K));
  if (has_clip) {
    l = (l < clip) ? l : clip;  // NaN compares false -> clip (evaluation.py:145-148)
    n = (n < clip) ? n : clip;
  }
  loss[b] = l;
  nll[b] = n;
}

}  // namespace

extern "C" int tv_adj_bernoulli_ll_fwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                       int64_t dim, const float *h, const float *adj, int biased,
                                       float alpha, float log_threshold, float *ll, void *stream) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0 && max_node_count >= 2, "tv_adj_bernoulli_ll_fwd: bad sizes");
  TV_REQUIRE(dim >= 1 && dim <= 80, "tv_adj_bernoulli_ll_fwd: dim must be in [1,80]: got %lld",
             (long long)dim);
  if (batch == 0) return TV_OK;
  TV_REQUIRE(h && adj && ll, "tv_adj_bernoulli_ll_fwd: null pointer");
  const size_t ldsb = sizeof(float) * 2 * dim * HS;
  hipLaunchKernelGGL(adj_ll_fwd_kernel, dim3((unsigned)(n_samples * batch)), dim3(256), ldsb,
                     (hipStream_t)stream, batch, max_node_count, (int)dim, h, adj, biased, alpha,
                     log_threshold, ll);
  TV_LAUNCH_CHECK("tv_adj_bernoulli_ll_fwd");
  return TV_OK;
}

extern "C" int tv_adj_bernoulli_ll_bwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                       int64_t dim, const float *h, const float *adj, int biased,
                                       float alpha, float log_threshold, const float *dll,
                                       float *dh, void *stream) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0 && max_node_count >= 2, "tv_adj_bernoulli_ll_bwd: bad sizes");
  TV_REQUIRE(dim >= 1 && dim <= 80, "tv_adj_bernoulli_ll_bwd: dim must be in [1,80]: got %lld",
             (long long)dim);
  if (batch == 0) return TV_OK;
  TV_REQUIRE(h && adj && dll && dh, "tv_adj_bernoulli_ll_bwd: null pointer");
  const size_t ldsb = sizeof(float) * (2 * dim * HS + 64 * 65);
  const dim3 grid((unsigned)(n_samples * batch)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dim <= 40)
    hipLaunchKernelGGL((adj_ll_bwd_kernel<10>), grid, block, ldsb, s, batch, max_node_count,
                       (int)dim, h, adj, biased, alpha, log_threshold, dll, dh);
  else
    hipLaunchKernelGGL((adj_ll_bwd_kernel<20>), grid, block, ldsb, s, batch, max_node_count,
                       (int)dim, h, adj, biased, alpha, log_threshold, dll, dh);
  TV_LAUNCH_CHECK("tv_adj_bernoulli_ll_bwd");
  return TV_OK;
}

static inline unsigned wave_row_blocks(int64_t rows) {
  int64_t b = tv_cdiv(rows, 4);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" int tv_categorical_ll_fwd(int64_t rows, int64_t classes, int64_t target_rows,
                                     const float *logits, const int64_t *target, int biased,
                                     float alpha, float log_threshold, float *ll, float *lse,
                                     void *stream) {
  TV_REQUIRE(rows >= 0 && classes >= 1 && target_rows >= 1, "tv_categorical_ll_fwd: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(rows % target_rows == 0, "tv_categorical_ll_fwd: rows must be a multiple of target_rows");
  TV_REQUIRE(logits && target && ll, "tv_categorical_ll_fwd: null pointer");
  hipLaunchKernelGGL(categorical_ll_fwd_kernel, dim3(wave_row_blocks(rows)), dim3(256), 0,
                     (hipStream_t)stream, rows, classes, target_rows, logits, target, biased, alpha,
                     log_threshold, ll, lse);
  TV_LAUNCH_CHECK("tv_categorical_ll_fwd");
  return TV_OK;
}

extern "C" int tv_categorical_ll_bwd(int64_t rows, int64_t classes, int64_t target_rows,
                                     const float *logits, const int64_t *target, int biased,
                                     float alpha, float log_threshold, const float *lse,
                                     const float *dll, float *dlogits, void *stream) {
  TV_REQUIRE(rows >= 0 && classes >= 1 && target_rows >= 1, "tv_categorical_ll_bwd: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(rows % target_rows == 0, "tv_categorical_ll_bwd: rows must be a multiple of target_rows");
  TV_REQUIRE(logits && target && lse && dll && dlogits, "tv_categorical_ll_bwd: null pointer");
  hipLaunchKernelGGL(categorical_ll_bwd_kernel, dim3(wave_row_blocks(rows)), dim3(256), 0,
                     (hipStream_t)stream, rows, classes, target_rows, logits, target, biased, alpha,
                     log_threshold, lse, dll, dlogits);
  TV_LAUNCH_CHECK("tv_categorical_ll_bwd");
  return TV_OK;
}

extern "C" int tv_normal_reparam_fwd(int64_t n_samples, int64_t batch, int64_t dim,
                                     const float *mean, const float *logstd_raw, float lo, float hi,
                                     const float *eps, float *z, float *logq, float *logstd,
                                     void *stream) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0 && dim >= 1, "tv_normal_reparam_fwd: bad sizes");
  if (batch == 0) return TV_OK;
  TV_REQUIRE(mean && logstd_raw && eps && z && logq && logstd, "tv_normal_reparam_fwd: null pointer");
  hipLaunchKernelGGL(normal_reparam_fwd_kernel, dim3((unsigned)tv_cdiv(n_samples * batch, 256)),
                     dim3(256), 0, (hipStream_t)stream, n_samples, batch, dim, mean, logstd_raw, lo,
                     hi, eps, z, logq, logstd);
  TV_LAUNCH_CHECK("tv_normal_reparam_fwd");
  return TV_OK;
}

extern "C" int tv_normal_reparam_bwd(int64_t n_samples, int64_t batch, int64_t dim,
                                     const float *logstd_raw, float lo, float hi, const float *eps,
                                     const float *dz, const float *dlogq, float *dmean,
                                     float *dlogstd_raw, void *stream) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0 && dim >= 1, "tv_normal_reparam_bwd: bad sizes");
  if (batch == 0) return TV_OK;
  TV_REQUIRE(logstd_raw && eps && dmean && dlogstd_raw, "tv_normal_reparam_bwd: null pointer");
  hipLaunchKernelGGL(normal_reparam_bwd_kernel, dim3((unsigned)tv_cdiv(batch * dim, 256)),
                     dim3(256), 0, (hipStream_t)stream, n_samples, batch, dim, logstd_raw, lo, hi,
                     eps, dz, dlogq, dmean, dlogstd_raw);
  TV_LAUNCH_CHECK("tv_normal_reparam_bwd");
  return TV_OK;
}

extern "C" int tv_normal_ll_fwd(int64_t rows, int64_t dim, const float *x, const float *mean,
                                const float *logstd_raw, float lo, float hi, float *ll,
                                void *stream) {
  TV_REQUIRE(rows >= 0 && dim >= 1, "tv_normal_ll_fwd: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && mean && logstd_raw && ll, "tv_normal_ll_fwd: null pointer");
  hipLaunchKernelGGL(normal_ll_fwd_kernel, dim3((unsigned)tv_cdiv(rows, 256)), dim3(256), 0,
                     (hipStream_t)stream, rows, dim, x, mean, logstd_raw, lo, hi, ll);
  TV_LAUNCH_CHECK("tv_normal_ll_fwd");
  return TV_OK;
}

extern "C" int tv_normal_ll_bwd(int64_t rows, int64_t dim, const float *x, const float *mean,
                                const float *logstd_raw, float lo, float hi, const float *dll,
                                float *dx, float *dmean, float *dlogstd_raw, void *stream) {
  TV_REQUIRE(rows >= 0 && dim >= 1, "tv_normal_ll_bwd: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(x && mean && logstd_raw && dll, "tv_normal_ll_bwd: null pointer");
  hipLaunchKernelGGL(normal_ll_bwd_kernel, dim3((unsigned)tv_cdiv(rows * dim, 256)), dim3(256), 0,
                     (hipStream_t)stream, rows, dim, x, mean, logstd_raw, lo, hi, dll, dx, dmean,
                     dlogstd_raw);
  TV_LAUNCH_CHECK("tv_normal_ll_bwd");
  return TV_OK;
}

static int fill_lat_args(LatArgs &a, const char *name, int64_t n_samples, int64_t batch,
                         int64_t max_node_count, const float *feats, const float *latency_pad,
                         const int64_t *node_count, const int64_t *node_type_pad,
                         const int64_t *node_offsets, const float *std_limit, int has_logstd_min,
                         float logstd_min, int mode, float std_threshold, float alpha,
                         float log_bias_threshold, int use_mask, int weighted) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0 && max_node_count >= 1, "%s: bad sizes", name);
  TV_REQUIRE(mode >= 0 && mode <= 3, "%s: mode must be 0..3", name);
  TV_REQUIRE(feats && latency_pad && node_count, "%s: null pointer", name);
  TV_REQUIRE(!std_limit || node_type_pad, "%s: std_limit needs node_type_pad", name);
  TV_REQUIRE(!node_offsets || use_mask, "%s: compact rows need the node-count mask", name);
  a.node_offsets = node_offsets;
  a.K = n_samples; a.B = batch; a.M = max_node_count; a.feats = feats;
  a.latency_pad = latency_pad; a.node_count = node_count; a.node_type_pad = node_type_pad;
  a.std_limit = std_limit; a.has_min = has_logstd_min; a.ls_min = logstd_min; a.mode = mode;
  a.thr = std_threshold; a.alpha = alpha; a.log_bias_thr = log_bias_threshold;
  a.use_mask = use_mask; a.weighted = weighted;
  return TV_OK;
}

extern "C" int tv_latency_ll_fwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                 const float *feats, const float *latency_pad,
                                 const int64_t *node_count, const int64_t *node_type_pad,
                                 const int64_t *node_offsets, const float *std_limit,
                                 int has_logstd_min, float logstd_min, int mode,
                                 float std_threshold, float alpha, float log_bias_threshold,
                                 int use_mask, int weighted, float *ll,
                                 float *std_out, void *stream) {
  LatArgs a;
  if (int rc = fill_lat_args(a, "tv_latency_ll_fwd", n_samples, batch, max_node_count, feats,
                             latency_pad, node_count, node_type_pad, node_offsets, std_limit, has_logstd_min,
                             logstd_min, mode, std_threshold, alpha, log_bias_threshold, use_mask,
                             weighted))
    return rc;
  if (batch == 0) return TV_OK;
  TV_REQUIRE(ll, "tv_latency_ll_fwd: null output");
  hipLaunchKernelGGL((latency_ll_kernel<false>), dim3(wave_row_blocks(n_samples * batch)),
                     dim3(256), 0, (hipStream_t)stream, a, ll, std_out, nullptr, nullptr);
  TV_LAUNCH_CHECK("tv_latency_ll_fwd");
  return TV_OK;
}

extern "C" int tv_latency_ll_bwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                 const float *feats, const float *latency_pad,
                                 const int64_t *node_count, const int64_t *node_type_pad,
                                 const int64_t *node_offsets, const float *std_limit,
                                 int has_logstd_min, float logstd_min, int mode,
                                 float std_threshold, float alpha, float log_bias_threshold,
                                 int use_mask, int weighted, const float *dll, float *dfeats,
                                 void *stream) {
  LatArgs a;
  if (int rc = fill_lat_args(a, "tv_latency_ll_bwd", n_samples, batch, max_node_count, feats,
                             latency_pad, node_count, node_type_pad, node_offsets, std_limit, has_logstd_min,
                             logstd_min, mode, std_threshold, alpha, log_bias_threshold, use_mask,
                             weighted))
    return rc;
  TV_REQUIRE(mode != 3, "tv_latency_ll_bwd: AnomalyDetectionNormal is a test-time likelihood");
  if (batch == 0) return TV_OK;
  TV_REQUIRE(dll && dfeats, "tv_latency_ll_bwd: null pointer");
  hipLaunchKernelGGL((latency_ll_kernel<true>), dim3(wave_row_blocks(n_samples * batch)), dim3(256),
                     0, (hipStream_t)stream, a, nullptr, nullptr, dll, dfeats);
  TV_LAUNCH_CHECK("tv_latency_ll_bwd");
  return TV_OK;
}

extern "C" int tv_score_reduce(int64_t n_samples, int64_t batch, const float *w, int has_clip,
                               float clip, float *loss, float *nll, void *stream) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0, "tv_score_reduce: bad sizes");
  if (batch == 0) return TV_OK;
  TV_REQUIRE(w && loss && nll, "tv_score_reduce: null pointer");
  hipLaunchKernelGGL(score_reduce_kernel, dim3((unsigned)tv_cdiv(batch, 256)), dim3(256), 0,
                     (hipStream_t)stream, n_samples, batch, w, has_clip, clip, loss, nll);
  TV_LAUNCH_CHECK("tv_score_reduce");
  return TV_OK;
}
