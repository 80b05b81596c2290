// RealNVP prior of the struct VAE: log p(z) under FlowDistribution(UnitNormal, flow.invert())
// (reference: tracegnn/models/trace_vae/model/realnvp_flow.py:18-49, struct_vae.py:148-149,
// 251-254).  `levels` x (ActNorm, affine coupling with one hidden layer and sigmoid scale,
// invertible dense), then the unit-normal density of the result.  z is [rows, D] with D = 10:
// the whole flow of a row fits one lane's registers, so forward and backward are one kernel each
// instead of ~300 tiny launches.
//
//   prepare : one lane per level - log|det K| and K^-T by Gauss-Jordan in float64 (parameter
//             only, shared by every row)
//   forward : one lane per row
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
__global__ void flow_prepare_kernel(int levels, int dim, FlowPtrs fp, float *ws) {
  const int l = threadIdx.x;
  if (l >= levels) return;
  constexpr int MD = TV_FLOW_MAX_DIM;
  double a[MD][2 * MD];
  const float *k = fp.p[l][8];
  for (int i = 0; i < dim; ++i)
    for (int j = 0; j < dim; ++j) {
      a[i][j] = (double)k[i * dim + j];
      a[i][dim + j] = i == j ? 1.0 : 0.0;
    }
  double logdet = 0.0;
  for (int c = 0; c < dim; ++c) {
    int piv = c;
    for (int r = c + 1; r < dim; ++r)
      if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 2 * dim; ++j) {
        const double t = a[c][j];
        a[c][j] = a[piv][j];
        a[piv][j] = t;
      }
    const double d = a[c][c];
    logdet += log(fabs(d));
    const double inv = 1.0 / d;
    for (int j = 0; j < 2 * dim; ++j) a[c][j] *= inv;
    for (int r = 0; r < dim; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      if (f != 0.0)
        for (int j = 0; j < 2 * dim; ++j) a[r][j] -= f * a[c][j];
    }
  }
  ws[l] = (float)logdet;
  float *inv_t = ws + levels + (int64_t)l * dim * dim;
  for (int i = 0; i < dim; ++i)
    for (int j = 0; j < dim; ++j) inv_t[i * dim + j] = (float)a[j][dim + i];   // (K^-1)^T
}

// one level forward on a row held in registers; returns the row's log-det contribution
template <int D, int H>
__device__ __forceinline__ float level_forward(const float *const *p, float slope, float (&x)[D]) {
  using X = Dims<D, H>;
  constexpr int N1 = X::N1, N2 = X::N2;
  const float *ab = p[0], *as = p[1], *w1 = p[2], *b1 = p[3], *ws = p[4], *bs = p[5], *wp = p[6],
              *bp = p[7], *kk = p[8];
#pragma unroll
  for (int j = 0; j < D; ++j) x[j] = (x[j] + ab[j]) * expf(as[j]);
  float sh[N2], pr[N2];
#pragma unroll
  for (int i = 0; i < N2; ++i) { sh[i] = bs[i]; pr[i] = bp[i]; }
  for (int k = 0; k < H; ++k) {
    float a = b1[k];
#pragma unroll
    for (int i = 0; i < N1; ++i) a = fmaf(w1[k * N1 + i], x[i], a);
    const float h = tv_lrelu(a, slope);
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      sh[i] = fmaf(ws[i * H + k], h, sh[i]);
      pr[i] = fmaf(wp[i * H + k], h, pr[i]);
    }
  }
  float ld = 0.f;
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    const float ls = -tv_softplus(-(pr[i] + 2.0f));     // logsigmoid(pre + 2)
    x[N1 + i] = (x[N1 + i] + sh[i]) * expf(ls);
    ld += ls;
  }
  float y[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) acc = fmaf(kk[i * D + j], x[j], acc);
    y[i] = acc;
  }
#pragma unroll
  for (int i = 0; i < D; ++i) x[i] = y[i];
  return ld;
}

__device__ __forceinline__ float flow_constant(int levels, int dim, const FlowPtrs &fp,
                                               const float *ws) {
  float c = 0.f;
  for (int l = 0; l < levels; ++l) {
    c += ws[l];
    const float *as = fp.p[l][1];
    for (int j = 0; j < dim; ++j) c += as[j];
  }
  return c;
}

template <int D, int H>
__global__ __launch_bounds__(64) void flow_fwd_kernel(int64_t rows, int levels, float slope,
                                                      const float *__restrict__ z, FlowPtrs fp,
                                                      const float *__restrict__ ws,
                                                      float *__restrict__ logp) {
  const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= rows) return;
  float x[D];
#pragma unroll
  for (int j = 0; j < D; ++j) x[j] = z[r * D + j];
  float ld = 0.f;
  for (int l = 0; l < levels; ++l) ld += level_forward<D, H>(fp.p[l], slope, x);
  float base = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) base += -0.5f * LOG_2PI - 0.5f * x[j] * x[j];
  logp[r] = base + ld + flow_constant(levels, D, fp, ws);
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
  constexpr int N1 = X::N1, N2 = X::N2, P = X::P;
  extern __shared__ float lds[];
  // staging (per level): h[H], da[H], dsh[N2], dpr[N2], u[D], dxn[D], dyb[D], dys[D]; then xs
  float *s_h = lds, *s_da = s_h + H * RS, *s_dsh = s_da + H * RS, *s_dpr = s_dsh + N2 * RS,
        *s_u = s_dpr + N2 * RS, *s_dxn = s_u + D * RS, *s_dyb = s_dxn + D * RS,
        *s_dys = s_dyb + D * RS, *s_xs = s_dys + D * RS;     // xs[levels][D][64]
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
    for (int l = 0; l < levels; ++l) {
#pragma unroll
      for (int j = 0; j < D; ++j) s_xs[(l * D + j) * 64 + t] = x[j];
      level_forward<D, H>(fp.p[l], slope, x);
    }
#pragma unroll
    for (int j = 0; j < D; ++j) dx[j] = -x[j] * g;
  }
  float *out = partial + (int64_t)blockIdx.x * ((int64_t)levels * P + 1);
  for (int l = levels - 1; l >= 0; --l) {
    const float *const *p = fp.p[l];
    if (lead) {
      const float *ab = p[0], *as = p[1], *w1 = p[2], *b1 = p[3], *ws = p[4], *bs = p[5],
                  *wp = p[6], *bp = p[7], *kk = p[8];
      float y[D], es[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        es[j] = expf(as[j]);
        y[j] = (s_xs[(l * D + j) * 64 + t] + ab[j]) * es[j];
      }
      float sh[N2], pr[N2];
#pragma unroll
      for (int i = 0; i < N2; ++i) { sh[i] = bs[i]; pr[i] = bp[i]; }
      for (int k = 0; k < H; ++k) {
        float a = b1[k];
#pragma unroll
        for (int i = 0; i < N1; ++i) a = fmaf(w1[k * N1 + i], y[i], a);
        const float h = tv_lrelu(a, slope);
        s_h[k * RS + t] = h;
#pragma unroll
        for (int i = 0; i < N2; ++i) {
          sh[i] = fmaf(ws[i * H + k], h, sh[i]);
          pr[i] = fmaf(wp[i * H + k], h, pr[i]);
        }
      }
      float sg[N2], u[D];
#pragma unroll
      for (int i = 0; i < N1; ++i) u[i] = y[i];
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        sg[i] = expf(-tv_softplus(-(pr[i] + 2.0f)));
        u[N1 + i] = (y[N1 + i] + sh[i]) * sg[i];
      }
      // invertible dense: xn = K u
      float du[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) acc = fmaf(kk[i * D + j], dx[i], acc);
        du[j] = acc;
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
      for (int k = 0; k < H; ++k) {
        float dh = 0.f;
#pragma unroll
        for (int i = 0; i < N2; ++i) {
          dh = fmaf(ws[i * H + k], dsh[i], dh);
          dh = fmaf(wp[i * H + k], dpr[i], dh);
        }
        const float da = s_h[k * RS + t] > 0.f ? dh : dh * slope;
        s_da[k * RS + t] = da;
#pragma unroll
        for (int i = 0; i < N1; ++i) dy[i] = fmaf(w1[k * N1 + i], da, dy[i]);
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
  return (int64_t)sizeof(float) * ((2 * H + 2 * X::N2 + 4 * D) * RS + (int64_t)levels * D * 64);
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
  hipLaunchKernelGGL(flow_prepare_kernel, dim3(1), dim3(64), 0, s, levels, D, fp, ws);
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
    static bool raised = false;          // per process and instantiation
    if (!raised) {
      TV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&flow_bwd_kernel<D, H>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised = true;
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
