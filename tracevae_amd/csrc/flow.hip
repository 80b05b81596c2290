// RealNVP prior of the struct VAE: log p(z) under FlowDistribution(UnitNormal, flow.invert())
// (reference: tracegnn/models/trace_vae/model/realnvp_flow.py:18-49, struct_vae.py:148-149,
// 251-254).  `levels` x (ActNorm, affine coupling with one hidden layer and sigmoid scale,
// invertible dense), then the unit-normal density of the result.  z is [rows, D] with D = 10:
// the whole flow of a row fits one lane's registers, so forward and backward are one kernel each
// instead of ~300 tiny launches.
//
//   prepare : one wave per level - log|det K| and K^-T by Gauss-Jordan in float64, a lane per
//             column (parameter only, shared by every row)
//   forward : one lane per row; a level's parameters are staged in LDS and read at
//             wave-uniform addresses
//   backward: a 256-lane workgroup per 64 rows.  Wave 0 recomputes the forward pass (level inputs
//             kept in LDS), walks the levels backwards and stages the per-row factors of every
//             parameter gradient in LDS; all four waves then sum the outer products over the 64
//             rows (one lane per parameter element) into a per-workgroup partial.  A second
//             kernel adds the partials in workgroup order (deterministic) and the log-det term.
#include "tv_common.h"

namespace {

constexpr int MAXL = TV_FLOW_MAX_LEVELS;
constexpr int NP = 9;   // parameter tensors per level, in the order of tracevae_hip.h
constexpr float LOG_2PI = 1.8378770664093453f;

struct FlowPtrs { const float *p[MAXL][NP]; };
struct FlowGradPtrs { float *g[MAXL][NP]; };

template <int D, int H>
struct Dims {
  static constexpr int N1 = D / 2, N2 = D - N1;
  // offsets of the tensors inside one level's packed gradient
  static constexpr int O_AB = 0, O_AS = D, O_W1 = 2 * D, O_B1 = O_W1 + H * N1,
                       O_WS = O_B1 + H, O_BS = O_WS + N2 * H, O_WP = O_BS + N2,
                       O_BP = O_WP + N2 * H, O_K = O_BP + N2, P = O_K + D * D;
};

__host__ __device__ inline int64_t flow_header_floats(int levels, int dim) {
  const int64_t n = (int64_t)levels * (1 + dim * dim);
  return (n + 3) / 4 * 4;
}

// ---------------------------------------------------------------------------------------
// one 64-lane workgroup per level, lane j owns column j of [K | I] (kept in LDS as float64)
__global__ __launch_bounds__(64) void flow_prepare_kernel(int levels, int dim, FlowPtrs fp,
                                                          float *ws) {
  constexpr int MD = TV_FLOW_MAX_DIM;
  __shared__ double a[MD][2 * MD + 1];
  __shared__ double colc[MD];
  __shared__ int s_piv;
  const int l = blockIdx.x, j = threadIdx.x, w = 2 * dim;
  const float *k = fp.p[l][8];
  if (j < w)
    for (int i = 0; i < dim; ++i)
      a[i][j] = j < dim ? (double)k[i * dim + j] : (j - dim == i ? 1.0 : 0.0);
  __syncthreads();
  double logdet = 0.0;
  for (int c = 0; c < dim; ++c) {
    if (j == 0) {
      int piv = c;
      for (int r = c + 1; r < dim; ++r)
        if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
      s_piv = piv;
    }
    __syncthreads();
    const int piv = s_piv;
    if (j < w && piv != c) {
      const double t = a[c][j];
      a[c][j] = a[piv][j];
      a[piv][j] = t;
    }
    __syncthreads();
    const double d = a[c][c];
    if (j < dim) colc[j] = a[j][c];
    __syncthreads();
    logdet += log(fabs(d));
    if (j < w) {
      const double pc = a[c][j] / d;
      for (int r = 0; r < dim; ++r)
        if (r != c) a[r][j] -= colc[r] * pc;
      a[c][j] = pc;
    }
    __syncthreads();
  }
  if (j == 0) ws[l] = (float)logdet;
  float *inv_t = ws + levels + (int64_t)l * dim * dim;
  if (j < dim)
    for (int i = 0; i < dim; ++i) inv_t[i * dim + j] = (float)a[j][dim + i];   // (K^-1)^T
}

// LDS image of one level's parameters.  The hidden layer is stored as one record per hidden
// unit k - {b1[k], w1[k][:], ws[:][k], wp[:][k]} padded to a multiple of 4 floats - so the k
// loops fetch a unit's 16 coefficients with four 16-byte LDS reads at a wave-uniform address.
template <int D, int H>
struct LdsW {
  using X = Dims<D, H>;
  static constexpr int RW = (1 + X::N1 + 2 * X::N2 + 3) / 4 * 4;
  static constexpr int L_REC = 0, L_AB = H * RW, L_AS = L_AB + D, L_BS = L_AS + D,
                       L_BP = L_BS + X::N2, L_K = (L_BP + X::N2 + 3) / 4 * 4,
                       SIZE = (L_K + D * D + 3) / 4 * 4;
  __device__ static constexpr int dest(int q, int e) {
    return q == 0 ? L_AB + e
         : q == 1 ? L_AS + e
         : q == 2 ? (e / X::N1) * RW + 1 + e % X::N1
         : q == 3 ? e * RW
         : q == 4 ? (e % H) * RW + 1 + X::N1 + e / H
         : q == 5 ? L_BS + e
         : q == 6 ? (e % H) * RW + 1 + X::N1 + X::N2 + e / H
         : q == 7 ? L_BP + e
                  : L_K + e;
  }
};

// every lane of the workgroup takes part; trip counts are compile-time, so every load of the
// level is issued before the first LDS store waits for it
template <int D, int H, int NT>
__device__ __forceinline__ void stage_level(const float *const *p, float *w, int t) {
  using X = Dims<D, H>;
  using L = LdsW<D, H>;
  constexpr int len[NP] = {D, D, H * X::N1, H, X::N2 * H, X::N2, X::N2 * H, X::N2, D * D};
  constexpr int TRIPS = (H * X::N2 + NT - 1) / NT;
  float v[NP][TRIPS];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const float *src = p[q];
#pragma unroll
    for (int i = 0; i < TRIPS; ++i)
      if (i * NT < len[q]) {
        const int e = i * NT + t;
        v[q][i] = e < len[q] ? src[e] : 0.f;
      }
  }
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int i = 0; i < TRIPS; ++i)
      if (i * NT < len[q]) {
        const int e = i * NT + t;
        if (e < len[q]) w[L::dest(q, e)] = v[q][i];
      }
}

template <int RW>
__device__ __forceinline__ void load_record(const float *w, int k, float (&r)[RW]) {
  const float4 *src = reinterpret_cast<const float4 *>(w + k * RW);
#pragma unroll
  for (int q = 0; q < RW / 4; ++q) {
    const float4 v = src[q];
    r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
  }
}

// exp / softplus on the hardware exp2 / log2 (the flow's terms are O(1); 1e-6 is plenty)
__device__ __forceinline__ float flow_softplus(float x) {
  return fmaxf(x, 0.f) + __logf(1.0f + __expf(-fabsf(x)));
}

// y = K x with K [D, D] read as 16-byte pieces
template <int D, bool TRANSPOSED>
__device__ __forceinline__ void dense_apply(const float *kk, const float (&x)[D], float (&y)[D]) {
  static_assert(D * D % 4 == 0, "dense kernel is read as float4");
#pragma unroll
  for (int i = 0; i < D; ++i) y[i] = 0.f;
  const float4 *k4 = reinterpret_cast<const float4 *>(kk);
#pragma unroll
  for (int q = 0; q < D * D / 4; ++q) {
    const float4 v = k4[q];
    const float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = 4 * q + j, row = e / D, col = e % D;
      if (TRANSPOSED) y[col] = fmaf(c[j], x[row], y[col]);
      else y[row] = fmaf(c[j], x[col], y[row]);
    }
  }
}

// one level forward on a row held in registers; returns the row's log-det contribution
template <int D, int H>
__device__ __forceinline__ float level_forward(const float *w, float slope, float (&x)[D]) {
  using X = Dims<D, H>;
  using L = LdsW<D, H>;
  constexpr int N1 = X::N1, N2 = X::N2, RW = L::RW;
  const float *ab = w + L::L_AB, *as = w + L::L_AS, *bs = w + L::L_BS, *bp = w + L::L_BP;
#pragma unroll
  for (int j = 0; j < D; ++j) x[j] = (x[j] + ab[j]) * __expf(as[j]);
  float sh[N2], pr[N2];
#pragma unroll
  for (int i = 0; i < N2; ++i) { sh[i] = bs[i]; pr[i] = bp[i]; }
#pragma unroll 4
  for (int k = 0; k < H; ++k) {
    float r[RW];
    load_record<RW>(w, k, r);
    float a = r[0];
#pragma unroll
    for (int i = 0; i < N1; ++i) a = fmaf(r[1 + i], x[i], a);
    const float h = tv_lrelu(a, slope);
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      sh[i] = fmaf(r[1 + N1 + i], h, sh[i]);
      pr[i] = fmaf(r[1 + N1 + N2 + i], h, pr[i]);
    }
  }
  float ld = 0.f;
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    const float ls = -flow_softplus(-(pr[i] + 2.0f));     // logsigmoid(pre + 2)
    x[N1 + i] = (x[N1 + i] + sh[i]) * __expf(ls);
    ld += ls;
  }
  float y[D];
  dense_apply<D, false>(w + L::L_K, x, y);
#pragma unroll
  for (int i = 0; i < D; ++i) x[i] = y[i];
  return ld;
}

template <int D, int H>
__global__ __launch_bounds__(64) void flow_fwd_kernel(int64_t rows, int levels, float slope,
                                                      const float *__restrict__ z, FlowPtrs fp,
                                                      const float *__restrict__ ws,
                                                      float *__restrict__ logp) {
  using X = Dims<D, H>;
  __shared__ __align__(16) float s_w[LdsW<D, H>::SIZE];
  const int t = threadIdx.x;
  const int64_t r = (int64_t)blockIdx.x * 64 + t;
  const bool valid = r < rows;
  float x[D];
#pragma unroll
  for (int j = 0; j < D; ++j) x[j] = valid ? z[r * D + j] : 0.f;
  float ld = 0.f;
  for (int l = 0; l < levels; ++l) {
    __syncthreads();
    stage_level<D, H, 64>(fp.p[l], s_w, t);
    __syncthreads();
    ld += level_forward<D, H>(s_w, slope, x) + ws[l];
#pragma unroll
    for (int j = 0; j < D; ++j) ld += s_w[LdsW<D, H>::L_AS + j];     // ActNorm log-det: sum(pre_scale)
  }
  float base = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) base += -0.5f * LOG_2PI - 0.5f * x[j] * x[j];
  if (valid) logp[r] = base + ld;
}

// ---------------------------------------------------------------------------------------
// data-dependent initialisation of the ActNorm layers (tensorkit ActNorm on its first batch,
// reference train.py:188-199: one forward pass on init_batch_size traces): level by level,
// bias = -mean and pre_scale = -log(max(std, 1e-6)) of the level's INPUT over the batch (biased
// variance, two passes in float64), then the level is applied with those values.  One
// workgroup, a lane per row (rows <= 1024): the levels are sequential and the batch is tiny.
template <int D, int H>
__global__ __launch_bounds__(1024) void flow_actnorm_init_kernel(int rows, int levels, float slope,
                                                                 const float *__restrict__ z,
                                                                 FlowPtrs fp, FlowGradPtrs out) {
  using L = LdsW<D, H>;
  __shared__ __align__(16) float s_w[L::SIZE];
  __shared__ double s_red[16][D];
  __shared__ double s_mean[D];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool valid = t < rows;
  float x[D];
#pragma unroll
  for (int j = 0; j < D; ++j) x[j] = valid ? z[(int64_t)t * D + j] : 0.f;
  for (int l = 0; l < levels; ++l) {
    __syncthreads();
    stage_level<D, H, 1024>(fp.p[l], s_w, t);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const double v = tv_wave_sum_d(valid ? (double)x[j] : 0.0);
      if (lane == 0) s_red[wave][j] = v;
    }
    __syncthreads();
    if (t < D) {
      double m = 0.0;
      for (int w = 0; w < 16; ++w) m += s_red[w][t];
      s_mean[t] = m / rows;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const double d = valid ? (double)x[j] - s_mean[j] : 0.0;
      const double v = tv_wave_sum_d(d * d);
      if (lane == 0) s_red[wave][j] = v;
    }
    __syncthreads();
    if (t < D) {
      double q = 0.0;
      for (int w = 0; w < 16; ++w) q += s_red[w][t];
      const float sd = (float)sqrt(q / rows);
      const float bias = (float)(-s_mean[t]);
      const float pre = -logf(fmaxf(sd, 1e-6f));
      out.g[l][0][t] = bias;
      out.g[l][1][t] = pre;
      s_w[L::L_AB + t] = bias;
      s_w[L::L_AS + t] = pre;
    }
    __syncthreads();
    level_forward<D, H>(s_w, slope, x);
  }
}

// The same initialisation for a batch of MORE than 1024 rows (train.init_batch_size above 1024,
// or a first log_prob call on K*B rows of a model that has not been initialised): one workgroup
// again - the levels are sequential and the statistics are over the whole batch - with every
// lane striding over rows t, t + 1024, ...  A lane cannot keep all its rows in registers, so
// the input of level l is recomputed from z through the (by then initialised) levels 0..l-1 for
// each of the two passes (mean, then squared deviations, float64 as above).  The chunk loops
// have the same trip count in every lane, so the barriers inside stage_level's callers line up.
// Work: levels^2 * rows level evaluations, a few milliseconds at 10^5 rows, once per model.
template <int D, int H>
__global__ __launch_bounds__(1024) void flow_actnorm_init_strided_kernel(int64_t rows, int levels,
                                                                         float slope,
                                                                         const float *__restrict__ z,
                                                                         FlowPtrs fp, FlowGradPtrs out) {
  using L = LdsW<D, H>;
  __shared__ __align__(16) float s_w[L::SIZE];
  __shared__ double s_red[16][D];
  __shared__ double s_mean[D];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t chunks = (rows + 1023) / 1024;
  for (int l = 0; l < levels; ++l) {
    for (int pass = 0; pass < 2; ++pass) {
      double acc[D];
#pragma unroll
      for (int j = 0; j < D; ++j) acc[j] = 0.0;
      for (int64_t c = 0; c < chunks; ++c) {
        const int64_t r = c * 1024 + t;
        const bool valid = r < rows;
        float x[D];
#pragma unroll
        for (int j = 0; j < D; ++j) x[j] = valid ? z[r * D + j] : 0.f;
        for (int l2 = 0; l2 < l; ++l2) {
          __syncthreads();
          stage_level<D, H, 1024>(fp.p[l2], s_w, t);      // reads the values written below
          __syncthreads();
          level_forward<D, H>(s_w, slope, x);
        }
        if (valid) {
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const double d = pass == 0 ? (double)x[j] : (double)x[j] - s_mean[j];
            acc[j] += pass == 0 ? d : d * d;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const double v = tv_wave_sum_d(acc[j]);
        if (lane == 0) s_red[wave][j] = v;
      }
      __syncthreads();
      if (t < D) {
        double m = 0.0;
        for (int w = 0; w < 16; ++w) m += s_red[w][t];
        if (pass == 0) {
          s_mean[t] = m / (double)rows;
        } else {
          const float sd = (float)sqrt(m / (double)rows);
          out.g[l][0][t] = (float)(-s_mean[t]);
          out.g[l][1][t] = -logf(fmaxf(sd, 1e-6f));
        }
      }
      __threadfence();          // the level's bias / pre_scale are read back from memory above
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------
constexpr int RS = 65;   // LDS row stride of the staging arrays (64 rows + 1: conflict-free)

template <int D, int H>
__global__ __launch_bounds__(256) void flow_bwd_kernel(int64_t rows, int levels, float slope,
                                                       const float *__restrict__ z, FlowPtrs fp,
                                                       const float *__restrict__ dlogp,
                                                       float *__restrict__ dz,
                                                       float *__restrict__ partial) {
  using X = Dims<D, H>;
  using L = LdsW<D, H>;
  constexpr int N1 = X::N1, N2 = X::N2, P = X::P, RW = L::RW;
  constexpr int STAGE_FLOATS = (2 * H + 2 * N2 + 4 * D) * RS;
  extern __shared__ __align__(16) float lds[];
  // staging (per level): h[H], da[H], dsh[N2], dpr[N2], u[D], dxn[D], dyb[D], dys[D]; then xs
  float *s_h = lds, *s_da = s_h + H * RS, *s_dsh = s_da + H * RS, *s_dpr = s_dsh + N2 * RS,
        *s_u = s_dpr + N2 * RS, *s_dxn = s_u + D * RS, *s_dyb = s_dxn + D * RS,
        *s_dys = s_dyb + D * RS,
        *s_w = lds + (STAGE_FLOATS + 3) / 4 * 4,             // LdsW image of a level
        *s_xs = s_w + L::SIZE;                               // xs[levels][D][64]
  const int t = threadIdx.x;
  const int64_t r = (int64_t)blockIdx.x * 64 + t;
  const bool lead = t < 64;
  const bool valid = lead && r < rows;
  float x[D], dx[D];
  float g = 0.f;
  if (lead) {
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] = valid ? z[r * D + j] : 0.f;
    g = valid ? dlogp[r] : 0.f;
  }
  for (int l = 0; l < levels; ++l) {
    __syncthreads();
    stage_level<D, H, 256>(fp.p[l], s_w, t);
    __syncthreads();
    if (lead) {
#pragma unroll
      for (int j = 0; j < D; ++j) s_xs[(l * D + j) * 64 + t] = x[j];
      level_forward<D, H>(s_w, slope, x);
    }
  }
  if (lead) {
#pragma unroll
    for (int j = 0; j < D; ++j) dx[j] = -x[j] * g;
  }
  float *out = partial + (int64_t)blockIdx.x * ((int64_t)levels * P + 1);
  for (int l = levels - 1; l >= 0; --l) {
    if (l != levels - 1) {             // the last level's parameters are still staged
      stage_level<D, H, 256>(fp.p[l], s_w, t);
      __syncthreads();
    }
    if (lead) {
      const float *ab = s_w + L::L_AB, *as = s_w + L::L_AS, *bs = s_w + L::L_BS,
                  *bp = s_w + L::L_BP, *kk = s_w + L::L_K;
      float y[D], es[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        es[j] = __expf(as[j]);
        y[j] = (s_xs[(l * D + j) * 64 + t] + ab[j]) * es[j];
      }
      float sh[N2], pr[N2];
#pragma unroll
      for (int i = 0; i < N2; ++i) { sh[i] = bs[i]; pr[i] = bp[i]; }
#pragma unroll 4
      for (int k = 0; k < H; ++k) {
        float r[RW];
        load_record<RW>(s_w, k, r);
        float a = r[0];
#pragma unroll
        for (int i = 0; i < N1; ++i) a = fmaf(r[1 + i], y[i], a);
        const float h = tv_lrelu(a, slope);
        s_h[k * RS + t] = h;
#pragma unroll
        for (int i = 0; i < N2; ++i) {
          sh[i] = fmaf(r[1 + N1 + i], h, sh[i]);
          pr[i] = fmaf(r[1 + N1 + N2 + i], h, pr[i]);
        }
      }
      float sg[N2], u[D];
#pragma unroll
      for (int i = 0; i < N1; ++i) u[i] = y[i];
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        sg[i] = __expf(-flow_softplus(-(pr[i] + 2.0f)));
        u[N1 + i] = (y[N1 + i] + sh[i]) * sg[i];
      }
      // invertible dense: xn = K u
      float du[D];
      dense_apply<D, true>(kk, dx, du);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        s_u[j * RS + t] = u[j];
        s_dxn[j * RS + t] = dx[j];
      }
      // coupling
      float dsh[N2], dpr[N2], dy[D];
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        const float dy2 = du[N1 + i];
        dsh[i] = dy2 * sg[i];
        dpr[i] = dy2 * (y[N1 + i] + sh[i]) * sg[i] * (1.f - sg[i]) + g * (1.f - sg[i]);
        dy[N1 + i] = dsh[i];
        s_dsh[i * RS + t] = dsh[i];
        s_dpr[i * RS + t] = dpr[i];
      }
#pragma unroll
      for (int i = 0; i < N1; ++i) dy[i] = du[i];
#pragma unroll 4
      for (int k = 0; k < H; ++k) {
        float dh = 0.f;
        float r[RW];
        load_record<RW>(s_w, k, r);
#pragma unroll
        for (int i = 0; i < N2; ++i) {
          dh = fmaf(r[1 + N1 + i], dsh[i], dh);
          dh = fmaf(r[1 + N1 + N2 + i], dpr[i], dh);
        }
        const float da = s_h[k * RS + t] > 0.f ? dh : dh * slope;
        s_da[k * RS + t] = da;
#pragma unroll
        for (int i = 0; i < N1; ++i) dy[i] = fmaf(r[1 + i], da, dy[i]);
      }
      // act-norm: y = (x + b) * exp(s); the constant sum(s) of log p adds g to every ds
#pragma unroll
      for (int j = 0; j < D; ++j) {
        dx[j] = dy[j] * es[j];
        s_dyb[j * RS + t] = dx[j];
        s_dys[j * RS + t] = fmaf(dy[j], y[j], g);
      }
    }
    __syncthreads();
    for (int o = t; o < P; o += 256) {
      const float *fa, *fb = nullptr;
      if (o < X::O_AS) fa = s_dyb + (o - X::O_AB) * RS;
      else if (o < X::O_W1) fa = s_dys + (o - X::O_AS) * RS;
      else if (o < X::O_B1) { const int e = o - X::O_W1; fa = s_da + (e / N1) * RS; fb = s_u + (e % N1) * RS; }
      else if (o < X::O_WS) fa = s_da + (o - X::O_B1) * RS;
      else if (o < X::O_BS) { const int e = o - X::O_WS; fa = s_dsh + (e / H) * RS; fb = s_h + (e % H) * RS; }
      else if (o < X::O_WP) fa = s_dsh + (o - X::O_BS) * RS;
      else if (o < X::O_BP) { const int e = o - X::O_WP; fa = s_dpr + (e / H) * RS; fb = s_h + (e % H) * RS; }
      else if (o < X::O_K) fa = s_dpr + (o - X::O_BP) * RS;
      else { const int e = o - X::O_K; fa = s_dxn + (e / D) * RS; fb = s_u + (e % D) * RS; }
      float acc = 0.f;
      if (fb) {
#pragma unroll 8
        for (int q = 0; q < 64; ++q) acc = fmaf(fa[q], fb[q], acc);
      } else {
#pragma unroll 8
        for (int q = 0; q < 64; ++q) acc += fa[q];
      }
      out[(int64_t)l * P + o] = acc;
    }
    __syncthreads();
  }
  if (lead) {
    if (valid) {
#pragma unroll
      for (int j = 0; j < D; ++j) dz[r * D + j] = dx[j];
    }
    const float gs = tv_wave_sum(g);
    if (t == 0) out[(int64_t)levels * P] = gs;
  }
}

template <int D, int H>
__global__ __launch_bounds__(256) void flow_grad_reduce_kernel(int levels, int64_t nblk,
                                                               const float *__restrict__ partial,
                                                               const float *__restrict__ ws,
                                                               FlowGradPtrs gp) {
  using X = Dims<D, H>;
  constexpr int P = X::P;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)levels * P) return;
  const int l = (int)(e / P), o = (int)(e % P);
  const int64_t slot = (int64_t)levels * P + 1;
  double acc = 0.0;
  for (int64_t b = 0; b < nblk; ++b) acc += (double)partial[b * slot + e];
  int which, idx;
  if (o < X::O_AS) { which = 0; idx = o; }
  else if (o < X::O_W1) { which = 1; idx = o - X::O_AS; }
  else if (o < X::O_B1) { which = 2; idx = o - X::O_W1; }
  else if (o < X::O_WS) { which = 3; idx = o - X::O_B1; }
  else if (o < X::O_BS) { which = 4; idx = o - X::O_WS; }
  else if (o < X::O_WP) { which = 5; idx = o - X::O_BS; }
  else if (o < X::O_BP) { which = 6; idx = o - X::O_WP; }
  else if (o < X::O_K) { which = 7; idx = o - X::O_BP; }
  else {
    which = 8;
    idx = o - X::O_K;
    double gsum = 0.0;                       // d log|det K| / dK = K^-T, once per row
    for (int64_t b = 0; b < nblk; ++b) gsum += (double)partial[b * slot + slot - 1];
    acc += gsum * (double)ws[levels + (int64_t)l * D * D + idx];
  }
  float *dst = gp.g[l][which];
  if (dst) dst[idx] = (float)acc;
}

template <int D, int H>
int64_t bwd_lds_bytes(int levels) {
  using X = Dims<D, H>;
  return (int64_t)sizeof(float) * (((2 * H + 2 * X::N2 + 4 * D) * RS + 3) / 4 * 4 +
                                   LdsW<D, H>::SIZE + (int64_t)levels * D * 64);
}

int check_dims(const char *who, int64_t rows, int levels, int dim, int hidden) {
  TV_REQUIRE(rows >= 0, "%s: rows < 0", who);
  TV_REQUIRE(levels >= 1 && levels <= MAXL, "%s: levels must be 1..%d, got %d", who, MAXL, levels);
  TV_REQUIRE(tv_realnvp_supported(dim, hidden), "%s: no kernel for dim %d / hidden %d", who, dim,
             hidden);
  return TV_OK;
}

int fill_ptrs(const char *who, int levels, const float *const *params, FlowPtrs *fp) {
  TV_REQUIRE(params, "%s: null parameter table", who);
  for (int l = 0; l < levels; ++l)
    for (int i = 0; i < NP; ++i) {
      TV_REQUIRE(params[l * NP + i], "%s: null parameter %d of level %d", who, i, l);
      fp->p[l][i] = params[l * NP + i];
    }
  return TV_OK;
}

template <int D, int H>
int run_fwd(int64_t rows, int levels, float slope, const float *z, const FlowPtrs &fp, float *ws,
            float *logp, hipStream_t s) {
  hipLaunchKernelGGL(flow_prepare_kernel, dim3(levels), dim3(64), 0, s, levels, D, fp, ws);
  TV_LAUNCH_CHECK("tv_realnvp_logprob_fwd(prepare)");
  if (rows == 0) return TV_OK;
  hipLaunchKernelGGL((flow_fwd_kernel<D, H>), dim3((unsigned)tv_cdiv(rows, 64)), dim3(64), 0, s, rows,
                     levels, slope, z, fp, ws, logp);
  TV_LAUNCH_CHECK("tv_realnvp_logprob_fwd");
  return TV_OK;
}

template <int D, int H>
int run_bwd(int64_t rows, int levels, float slope, const float *z, const FlowPtrs &fp,
            const float *dlogp, float *ws, float *dz, const FlowGradPtrs &gp, hipStream_t s) {
  const int64_t nblk = tv_cdiv(rows, 64);
  float *partial = ws + flow_header_floats(levels, D);
  const int64_t lds = bwd_lds_bytes<D, H>(levels);
  TV_REQUIRE(lds <= 160 * 1024, "tv_realnvp_logprob_bwd: %lld bytes of LDS needed", (long long)lds);
  if (lds > 64 * 1024) {
    static bool raised[64] = {};         // per device and instantiation
    int device = 0;
    TV_HIP(hipGetDevice(&device));
    if (!raised[device & 63]) {
      TV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&flow_bwd_kernel<D, H>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised[device & 63] = true;
    }
  }
  if (nblk > 0) {
    hipLaunchKernelGGL((flow_bwd_kernel<D, H>), dim3((unsigned)nblk), dim3(256), (size_t)lds, s, rows,
                       levels, slope, z, fp, dlogp, dz, partial);
    TV_LAUNCH_CHECK("tv_realnvp_logprob_bwd");
  }
  hipLaunchKernelGGL((flow_grad_reduce_kernel<D, H>),
                     dim3((unsigned)tv_cdiv((int64_t)levels * Dims<D, H>::P, 256)), dim3(256), 0, s,
                     levels, nblk, partial, ws, gp);
  TV_LAUNCH_CHECK("tv_realnvp_logprob_bwd(reduce)");
  return TV_OK;
}

template <int D, int H>
int run_actnorm_init(int64_t rows, int levels, float slope, const float *z, const FlowPtrs &fp,
                     const FlowGradPtrs &out, hipStream_t s) {
  if (rows <= 1024)
    hipLaunchKernelGGL((flow_actnorm_init_kernel<D, H>), dim3(1), dim3(1024), 0, s, (int)rows, levels,
                       slope, z, fp, out);
  else
    hipLaunchKernelGGL((flow_actnorm_init_strided_kernel<D, H>), dim3(1), dim3(1024), 0, s, rows,
                       levels, slope, z, fp, out);
  TV_LAUNCH_CHECK("tv_realnvp_actnorm_init");
  return TV_OK;
}

}  // namespace

extern "C" int tv_realnvp_supported(int dim, int hidden) {
  return (dim == 10 || dim == 16 || dim == 20) && hidden == 64;
}

extern "C" int64_t tv_realnvp_workspace_floats(int64_t rows, int levels, int dim, int hidden) {
  if (rows < 0 || levels < 1 || levels > MAXL || !tv_realnvp_supported(dim, hidden)) return -1;
  const int n1 = dim / 2, n2 = dim - n1;
  const int64_t p = 2 * dim + hidden * n1 + hidden + 2 * (n2 * hidden + n2) + dim * dim;
  return flow_header_floats(levels, dim) + tv_cdiv(rows, 64) * ((int64_t)levels * p + 1);
}

#define TV_FLOW_DISPATCH(fn, ...)                                  \
  do {                                                             \
    if (dim == 10) return fn<10, 64>(__VA_ARGS__);                 \
    if (dim == 16) return fn<16, 64>(__VA_ARGS__);                 \
    return fn<20, 64>(__VA_ARGS__);                                \
  } while (0)

extern "C" int tv_realnvp_logprob_fwd(int64_t rows, int levels, int dim, int hidden, float slope,
                                      const float *z, const float *const *params, float *ws,
                                      float *log_prob, void *stream) {
  if (int rc = check_dims("tv_realnvp_logprob_fwd", rows, levels, dim, hidden)) return rc;
  TV_REQUIRE(ws && (rows == 0 || (z && log_prob)), "tv_realnvp_logprob_fwd: null pointer");
  FlowPtrs fp{};
  if (int rc = fill_ptrs("tv_realnvp_logprob_fwd", levels, params, &fp)) return rc;
  hipStream_t s = (hipStream_t)stream;
  TV_FLOW_DISPATCH(run_fwd, rows, levels, slope, z, fp, ws, log_prob, s);
}

extern "C" int tv_realnvp_logprob_bwd(int64_t rows, int levels, int dim, int hidden, float slope,
                                      const float *z, const float *const *params,
                                      const float *dlog_prob, float *ws, float *dz,
                                      float *const *grads, void *stream) {
  if (int rc = check_dims("tv_realnvp_logprob_bwd", rows, levels, dim, hidden)) return rc;
  TV_REQUIRE(ws && grads && (rows == 0 || (z && dlog_prob && dz)),
             "tv_realnvp_logprob_bwd: null pointer");
  FlowPtrs fp{};
  if (int rc = fill_ptrs("tv_realnvp_logprob_bwd", levels, params, &fp)) return rc;
  FlowGradPtrs gp{};
  for (int l = 0; l < levels; ++l)
    for (int i = 0; i < NP; ++i) gp.g[l][i] = grads[l * NP + i];
  hipStream_t s = (hipStream_t)stream;
  TV_FLOW_DISPATCH(run_bwd, rows, levels, slope, z, fp, dlog_prob, ws, dz, gp, s);
}

extern "C" int tv_realnvp_actnorm_init(int64_t rows, int levels, int dim, int hidden, float slope,
                                       const float *z, float *const *params, void *stream) {
  if (int rc = check_dims("tv_realnvp_actnorm_init", rows, levels, dim, hidden)) return rc;
  TV_REQUIRE(rows >= 1 && rows <= (1 << 24),
             "tv_realnvp_actnorm_init: the initialisation batch must hold 1..2^24 rows, got %lld",
             (long long)rows);
  TV_REQUIRE(z, "tv_realnvp_actnorm_init: null pointer");
  FlowPtrs fp{};
  if (int rc = fill_ptrs("tv_realnvp_actnorm_init", levels, params, &fp)) return rc;
  FlowGradPtrs out{};
  for (int l = 0; l < levels; ++l) {
    out.g[l][0] = params[l * NP + 0];
    out.g[l][1] = params[l * NP + 1];
  }
  hipStream_t s = (hipStream_t)stream;
  TV_FLOW_DISPATCH(run_actnorm_init, rows, levels, slope, z, fp, out, s);
}
