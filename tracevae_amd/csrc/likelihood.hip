// (c)/(d) likelihood kernels: adjacency-logit contraction fused with the Bernoulli
// log-likelihood, categorical / normal log-likelihoods with the reference's Biased / Safe /
// Masked variants, reparameterised sampling, and the K-sample score reduction.
// All sums are fixed-order (per-thread partials -> wave / workgroup tree), no atomics.
#include "tv_common.h"

#include <mutex>

namespace {

constexpr float LOG_2PI = 1.8378770664093453f;

// ------------------------------------------------------------------------------------
// p(adj | z) on the matrix cores: l_ij = <h_i, h_j> for i < j, Bernoulli log-likelihood over
// all T = M(M-1)/2 pairs (tensor_utils.py:54-67, struct_vae.py:281-294).  [K,B,M,M] is never
// materialised: a wave owns one 32x32 tile of node pairs, forms its logits with a chain of
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) and applies softplus / the biased rule to the 16
// logits each lane then holds.
//
// Operand layout.  A 32x32x2 step multiplies A[i = lane&31][k = lane>>5] by
// B[k = lane>>5][j = lane&31].  The contraction index is split in two halves of Dh = ceil(D/2):
// lanes 0-31 walk k = 0 .. Dh-1, lanes 32-63 walk k = Dh .. D-1, so a lane reads one contiguous
// half of its h row (float4 loads when D % 8 == 0) and step s pairs k = s with k = Dh + s, the
// same for both operands: a fixed summation order.  Steps beyond Dh multiply zeros.
// C/D layout: lane holds column lane&31, rows (r&3) + 8(r>>2) + 4(lane>>5) in register r.
//
// Grid: forward  = (sample, trace) x groups of four tiles of the upper block triangle; each
//                  workgroup writes ONE partial sum, a second kernel adds a trace's partials
//                  in order (float64) - fixed order, no atomics;
//       backward = (sample, trace) x 32-row block I; the four waves of a workgroup split the
//                  column blocks J, each forms X = G_JI (j in registers, i on the lane), which
//                  is at once the A operand of dH_I += X^T H_J (the sum runs over X's row
//                  index, so no lane movement), and the four partial dH_I tiles are added in
//                  wave order through LDS.
// adj is read in its coalesced orientation (lane = the larger node index) into a 32x33 LDS
// tile per wave and picked up from there in whatever orientation the tile needs.
// Algorithmic bytes per (k,b): 4*M*D (h) + 4*T (adj, L2-resident across the K samples).
// ------------------------------------------------------------------------------------
constexpr int AT = 32;                 // pair-tile edge
constexpr int ATS = AT + 1;            // LDS row stride of the adj tile
constexpr int ADJ_WAVES = 4;           // waves per workgroup
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int triu_index32(int i, int j, int M) {   // i < j < M <= 1536
  return i * (2 * M - i - 1) / 2 + (j - i - 1);
}
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Bernoulli terms of one logit with ONE exponential: e = exp(-|l|),
// softplus(l) = max(l, 0) + log(1 + e), sigmoid(l) = l >= 0 ? 1/(1+e) : e/(1+e).
// v_exp_f32 / v_log_f32 / v_rcp_f32 forms: 1 + e lies in (1, 2], where the hardware log is good
// to ~1e-7 absolute - the terms are summed next to |l|-sized values.
__device__ __forceinline__ float adj_log_prob(float x, float l, float e) {
  return x * l - (fmaxf(l, 0.f) + __logf(1.0f + e));
}
__device__ __forceinline__ float adj_sigmoid(float l, float e) {
  const float r = __frcp_rn(1.0f + e);
  return l >= 0.f ? r : e * r;
}

// this lane's half of row `row` of h: a[s] = h[row][half*Dh + s], zero beyond the half / D
template <int NV, bool VEC>
__device__ __forceinline__ void load_h_half(float (&a)[4 * NV], const float *__restrict__ hb,
                                            int row, int D, int Dh, int half) {
  const float *p = hb + (int64_t)row * D + half * Dh;
  if (VEC) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (4 * q < Dh) v = *reinterpret_cast<const float4 *>(p + 4 * q);
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
  } else {
    const int valid = min(Dh, D - half * Dh);
#pragma unroll
    for (int s = 0; s < 4 * NV; ++s) a[s] = (s < valid) ? p[s] : 0.f;
  }
}

// the 32x32 adj tile of (row block rb <= column block cb) into S[r][c]: entries with global
// row < column < M, zero elsewhere; lanes run along the column (coalesced 128-byte rows)
__device__ __forceinline__ void stage_adj_tile(float *S, const float *__restrict__ ab, int rb,
                                               int cb, int M, int lane) {
  const int c = lane & 31, half = lane >> 5, gc = cb * AT + c;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int r = 2 * t + half, gr = rb * AT + r;
    S[r * ATS + c] = (gr < gc && gc < M) ? ab[triu_index32(gr, gc, M)] : 0.f;
  }
}

template <int NV, bool VEC>
__global__ __launch_bounds__(64 * ADJ_WAVES) void adj_ll_fwd_kernel(
    int64_t B, int M, int D, int nb, int ntiles, int ngroups, const float *__restrict__ h,
    const float *__restrict__ adj, int biased, float alpha, float log_thr,
    float *__restrict__ partial) {
  __shared__ float red[ADJ_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
  const int64_t kb = blockIdx.x / ngroups;
  const int group = (int)(blockIdx.x - kb * ngroups);
  const float *hb = h + kb * M * D;
  const float *ab = adj + (kb % B) * ((int64_t)M * (M - 1) / 2);
  const int Dh = (D + 1) >> 1;
  float acc = 0.f;
  int t = group * ADJ_WAVES + wave;
  if (t < ntiles) {
    int rb = 0;                                   // tile t of the upper block triangle, row-major
    while (t >= nb - rb) { t -= nb - rb; ++rb; }
    const int cb = rb + t;
    float a[4 * NV], b[4 * NV];
    load_h_half<NV, VEC>(a, hb, min(rb * AT + col, M - 1), D, Dh, half);
    load_h_half<NV, VEC>(b, hb, min(cb * AT + col, M - 1), D, Dh, half);
    f32x16 l = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4 * NV; ++s) l = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], l, 0, 0, 0);
    const int gj = cb * AT + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gi = rb * AT + acc_row(r, half);
      if (gi < gj && gj < M) {
        const float x = ab[triu_index32(gi, gj, M)];
        float lp = adj_log_prob(x, l[r], __expf(-fabsf(l[r])));
        if (biased && lp < log_thr) lp *= alpha;
        acc += lp;
      }
    }
  }
  acc = tv_wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = red[0];
#pragma unroll
    for (int w = 1; w < ADJ_WAVES; ++w) s += red[w];
    partial[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(256) void adj_ll_reduce_kernel(int64_t KB, int ngroups,
                                                            const float *__restrict__ partial,
                                                            float *__restrict__ ll) {
  const int64_t kb = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (kb >= KB) return;
  double s = 0.0;
  for (int g = 0; g < ngroups; ++g) s += (double)partial[kb * ngroups + g];
  ll[kb] = (float)s;
}

// dh_i = sum_{j != i} g_ij h_j,  g_ij = dll * f'(lp_ij) * (adj_ij - sigmoid(l_ij))
// `jsplit` > 1 (few traces of many nodes): the column blocks of one row block are dealt to
// jsplit workgroups, each writes its partial dH_I tile to `out` = the workspace laid out
// [K*B][jsplit][M][D], and adj_dh_reduce_kernel adds the jsplit partials in order.
template <int NV, bool VEC, int ND>     // ND = ceil(D / 32) output column blocks
__global__ __launch_bounds__(64 * ADJ_WAVES) void adj_ll_bwd_kernel(
    int64_t B, int M, int D, int nb, int jsplit, const float *__restrict__ h,
    const float *__restrict__ adj, int biased, float alpha, float log_thr,
    const float *__restrict__ dll, float *__restrict__ out) {
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
  float *S = lds + wave * (AT * ATS);                          // this wave's adj tile
  float *sums = lds + ADJ_WAVES * (AT * ATS);                  // [ADJ_WAVES-1][ND*16][64]
  const int js = (int)(blockIdx.x % jsplit);
  const int64_t tile = blockIdx.x / jsplit, kb = tile / nb;
  const int ib = (int)(tile - kb * nb);
  const float *hb = h + kb * M * D;
  const float *ab = adj + (kb % B) * ((int64_t)M * (M - 1) / 2);
  const float dl = dll[kb];
  const int Dh = (D + 1) >> 1;
  const int gi = ib * AT + col;                                // this lane's node i
  float bi[4 * NV];
  load_h_half<NV, VEC>(bi, hb, min(gi, M - 1), D, Dh, half);
  f32x16 acc[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  const int stride = ADJ_WAVES * jsplit;
  const int rounds = (nb + stride - 1) / stride;               // every wave meets every barrier
  for (int it = 0; it < rounds; ++it) {
    const int jb = (it * jsplit + js) * ADJ_WAVES + wave;
    const bool live = jb < nb;
    f32x16 x = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
      float aj[4 * NV];
      load_h_half<NV, VEC>(aj, hb, min(jb * AT + col, M - 1), D, Dh, half);
#pragma unroll
      for (int s = 0; s < 4 * NV; ++s) x = __builtin_amdgcn_mfma_f32_32x32x2f32(aj[s], bi[s], x, 0, 0, 0);
      stage_adj_tile(S, ab, min(ib, jb), max(ib, jb), M, lane);
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jl = acc_row(r, half), gj = jb * AT + jl;
        // S is [smaller node][larger node] of the block pair
        int sr, sc;
        if (jb > ib) { sr = col; sc = jl; }
        else if (jb < ib) { sr = jl; sc = col; }
        else { sr = min(col, jl); sc = max(col, jl); }
        const float av = S[sr * ATS + sc];
        float g = 0.f;
        if (gi != gj && gi < M && gj < M) {
          const float lv = x[r];
          const float e = __expf(-fabsf(lv));
          g = dl * (av - adj_sigmoid(lv, e));
          if (biased && adj_log_prob(av, lv, e) < log_thr) g *= alpha;
        }
        x[r] = g;
      }
      // dH_I += X^T H_J: A = X as it stands (i on the lane, j = the register's row), B = H_J
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const int dc = d * AT + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gj = min(jb * AT + acc_row(r, half), M - 1);
          const float hv = (dc < D) ? hb[(int64_t)gj * D + dc] : 0.f;
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[r], hv, acc[d], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // the four partial tiles, added in wave order
  if (wave > 0) {
    float *dst = sums + (wave - 1) * (ND * 16 * 64);
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(d * 16 + r) * 64 + lane] = acc[d][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const int dc = d * AT + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[d][r];
#pragma unroll
        for (int w = 1; w < ADJ_WAVES; ++w) v += sums[(w - 1) * (ND * 16 * 64) + (d * 16 + r) * 64 + lane];
        const int row = ib * AT + acc_row(r, half);
        if (row < M && dc < D) out[((kb * jsplit + js) * M + row) * D + dc] = v;
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
    // eight loads in flight per lane (one load per trip behind the branchy update streamed the
    // 5 GB of K = 100 scoring logits at 2.6 TB/s)
    for (int64_t c0 = lane; c0 < V; c0 += 64 * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (c0 + 64 * k < V) ? x[c0 + 64 * k] : -INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (v[k] > m) { s = s * expf(m - v[k]) + 1.0f; m = v[k]; }
        else if (v[k] != -INFINITY) { s += expf(v[k] - m); }
      }
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
    const float *__restrict__ lse, const float *__restrict__ dll, float *__restrict__ dlogits,
    int64_t ldd) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float *x = logits + r * V;
    float *dx = dlogits + r * ldd;
    int64_t t = target[r % target_rows];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const float ls = lse[r];
    float g = dll[r];
    if (biased && (x[t] - ls) < log_thr) g *= alpha;
    for (int64_t c = lane; c < V; c += 64)
      dx[c] = g * ((c == t ? 1.0f : 0.0f) - expf(x[c] - ls));
    if (V + lane < ldd) dx[V + lane] = 0.f;            // the pad columns of the row (fewer than 64)
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

__global__ __launch_bounds__(256) void adj_dh_reduce_kernel(int64_t KB, int64_t MD, int jsplit,
                                                            const float *__restrict__ part,
                                                            float *__restrict__ dh) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= KB * MD) return;
  const int64_t kb = e / MD, r = e - kb * MD;
  float s = part[(kb * jsplit) * MD + r];
  for (int j = 1; j < jsplit; ++j) s += part[(kb * jsplit + j) * MD + r];
  dh[e] = s;
}

// tiles and workgroups of the forward pass
static inline void adj_geometry(int64_t M, int *nb, int *ntiles, int *ngroups) {
  *nb = (int)tv_cdiv(M, AT);
  *ntiles = *nb * (*nb + 1) / 2;
  *ngroups = (*ntiles + ADJ_WAVES - 1) / ADJ_WAVES;
}

extern "C" int64_t tv_adj_bernoulli_ll_fwd_workspace_bytes(int64_t n_samples, int64_t batch,
                                                           int64_t max_node_count) {
  if (n_samples < 1 || batch < 0 || max_node_count < 2) return 0;
  int nb, ntiles, ngroups;
  adj_geometry(max_node_count, &nb, &ntiles, &ngroups);
  return (int64_t)sizeof(float) * n_samples * batch * ngroups;
}

static int adj_check(const char *who, int64_t n_samples, int64_t batch, int64_t max_node_count,
                     int64_t dim) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0 && max_node_count >= 2, "%s: bad sizes", who);
  TV_REQUIRE(max_node_count <= 1536, "%s: max_node_count must be <= 1536: got %lld", who,
             (long long)max_node_count);
  TV_REQUIRE(dim >= 1 && dim <= 80, "%s: dim must be in [1,80]: got %lld", who, (long long)dim);
  int nb, ntiles, ngroups;
  adj_geometry(max_node_count, &nb, &ntiles, &ngroups);
  TV_REQUIRE(n_samples * batch * (int64_t)(ngroups > nb ? ngroups : nb) < (int64_t)1 << 31,
             "%s: n_samples * batch too large for one launch", who);
  return TV_OK;
}

// the two launches of the forward pass; `partial` holds n_samples * batch * ngroups floats
static int adj_fwd_launch(int64_t n_samples, int64_t batch, int64_t M, int64_t dim, const float *h,
                          const float *adj, int biased, float alpha, float log_threshold,
                          float *partial, float *ll, hipStream_t s) {
  int nb, ntiles, ngroups;
  adj_geometry(M, &nb, &ntiles, &ngroups);
  const int64_t KB = n_samples * batch;
  const dim3 grid((unsigned)(KB * ngroups)), block(64 * ADJ_WAVES);
  const bool vec = dim % 8 == 0;
#define ADJ_FWD(NV, VEC)                                                                        \
  hipLaunchKernelGGL((adj_ll_fwd_kernel<NV, VEC>), grid, block, 0, s, batch, (int)M, (int)dim,  \
                     nb, ntiles, ngroups, h, adj, biased, alpha, log_threshold, partial)
  if (dim <= 40) { if (vec) ADJ_FWD(5, true); else ADJ_FWD(5, false); }
  else { if (vec) ADJ_FWD(10, true); else ADJ_FWD(10, false); }
#undef ADJ_FWD
  TV_LAUNCH_CHECK("tv_adj_bernoulli_ll_fwd");
  hipLaunchKernelGGL(adj_ll_reduce_kernel, dim3((unsigned)tv_cdiv(KB, 256)), dim3(256), 0, s, KB,
                     ngroups, partial, ll);
  TV_LAUNCH_CHECK("tv_adj_bernoulli_ll_fwd (reduce)");
  return TV_OK;
}

// ------------------------------------------------------------------------------------
// tv_edge_logits_triu: the logits themselves, for the callers that need them as a tensor (the
// prior-sampling path draws the adjacency from them; `.distribution.logits` of the adj node).
// One thread per pair (i < j) of a graph, in dense_to_triu order; fp32 fma chain over dim.
// ------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void edge_logits_triu_kernel(int64_t graphs, int M, int D,
                                                               const float *__restrict__ h,
                                                               float *__restrict__ out) {
  const int64_t T = (int64_t)M * (M - 1) / 2;
  const int64_t g = blockIdx.y;
  const float *hg = h + g * (int64_t)M * D;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T;
       t += (int64_t)gridDim.x * blockDim.x) {
    // row i of pair t: the largest i with i*(2M-i-1)/2 <= t
    int i = (int)((2.0 * M - 1.0 - sqrt((2.0 * M - 1.0) * (2.0 * M - 1.0) - 8.0 * (double)t)) * 0.5);
    while (i > 0 && (int64_t)i * (2 * M - i - 1) / 2 > t) --i;
    while ((int64_t)(i + 1) * (2 * M - i - 2) / 2 <= t) ++i;
    const int j = (int)(t - (int64_t)i * (2 * M - i - 1) / 2) + i + 1;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = fmaf(hg[(int64_t)i * D + d], hg[(int64_t)j * D + d], acc);
    out[g * T + t] = acc;
  }
}
}  // namespace

extern "C" int tv_edge_logits_triu(int64_t graphs, int64_t max_node_count, int64_t dim,
                                   const float *h, float *logits, void *stream) {
  TV_REQUIRE(graphs >= 0 && graphs <= 65535, "tv_edge_logits_triu: graphs must be in [0, 65535]: got %lld",
             (long long)graphs);
  TV_REQUIRE(max_node_count >= 2 && max_node_count <= 1536 && dim >= 1,
             "tv_edge_logits_triu: bad sizes (max_node_count in [2, 1536], dim >= 1)");
  if (graphs == 0) return TV_OK;
  TV_REQUIRE(h && logits, "tv_edge_logits_triu: null pointer");
  const int64_t T = max_node_count * (max_node_count - 1) / 2;
  int64_t bx = tv_cdiv(T, 256);
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(edge_logits_triu_kernel, dim3((unsigned)bx, (unsigned)graphs), dim3(256), 0,
                     (hipStream_t)stream, graphs, (int)max_node_count, (int)dim, h, logits);
  TV_LAUNCH_CHECK("tv_edge_logits_triu");
  return TV_OK;
}

extern "C" int tv_adj_bernoulli_ll_fwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                       int64_t dim, const float *h, const float *adj, int biased,
                                       float alpha, float log_threshold, float *ll,
                                       void *workspace, int64_t workspace_bytes, void *stream) {
  const int rc = adj_check("tv_adj_bernoulli_ll_fwd", n_samples, batch, max_node_count, dim);
  if (rc != TV_OK) return rc;
  if (batch == 0) return TV_OK;
  TV_REQUIRE(h && adj && ll, "tv_adj_bernoulli_ll_fwd: null pointer");
  const int64_t need = tv_adj_bernoulli_ll_fwd_workspace_bytes(n_samples, batch, max_node_count);
  if (!workspace || workspace_bytes < need) {
    tv_set_error("tv_adj_bernoulli_ll_fwd: workspace of %lld bytes needed, got %lld",
                 (long long)need, (long long)workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  return adj_fwd_launch(n_samples, batch, max_node_count, dim, h, adj, biased, alpha,
                        log_threshold, (float *)workspace, ll, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------
// tv_adj_bernoulli_ll_fwd_static: the same forward pass for callers that cannot allocate (HIP
// graph capture, tight serving loops).  The per-workgroup partial sums go to a file-scope
// device array instead of a caller workspace.  The array's NAME is a host-side stand-in, not a
// device address: every GPU holds its own copy at its own address, looked up once per device
// with hipGetSymbolAddress and kept; kernels only ever see that looked-up pointer.
// Every call on a GPU shares that GPU's array: the forward kernel overwrites every partial the
// reduce kernel of the same call reads (nothing has to start at zero), and calls that overlap
// on one GPU (two streams) race.
// ------------------------------------------------------------------------------------
static __device__ __align__(256) float scratch[TV_ADJ_STATIC_MAX_PARTIALS];

constexpr int ADJ_MAX_DEVICES = 64;
static std::mutex adj_scratch_mutex;
static float *adj_scratch_ptr[ADJ_MAX_DEVICES];     // per device, filled on first use

static int adj_static_scratch(float **out) {
  int dev = -1;
  TV_HIP(hipGetDevice(&dev));
  TV_REQUIRE(dev >= 0 && dev < ADJ_MAX_DEVICES, "tv_adj_bernoulli_ll_fwd_static: device %d", dev);
  std::lock_guard<std::mutex> guard(adj_scratch_mutex);
  if (adj_scratch_ptr[dev] == nullptr) {
    void *p = nullptr;
    TV_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));      // with device `dev` current
    TV_REQUIRE(p != nullptr, "tv_adj_bernoulli_ll_fwd_static: no address for the scratch array");
    adj_scratch_ptr[dev] = (float *)p;
  }
  *out = adj_scratch_ptr[dev];
  return TV_OK;
}

extern "C" int tv_adj_bernoulli_ll_fwd_static(int64_t n_samples, int64_t batch,
                                              int64_t max_node_count, int64_t dim,
                                              const float *h, const float *adj, int biased,
                                              float alpha, float log_threshold, float *ll,
                                              void *stream) {
  int rc = adj_check("tv_adj_bernoulli_ll_fwd_static", n_samples, batch, max_node_count, dim);
  if (rc != TV_OK) return rc;
  if (batch == 0) return TV_OK;
  TV_REQUIRE(h && adj && ll, "tv_adj_bernoulli_ll_fwd_static: null pointer");
  const int64_t need =
      tv_adj_bernoulli_ll_fwd_workspace_bytes(n_samples, batch, max_node_count) / (int64_t)sizeof(float);
  if (need > TV_ADJ_STATIC_MAX_PARTIALS) {
    tv_set_error("tv_adj_bernoulli_ll_fwd_static: n_samples * batch * tile groups = %lld partial "
                 "sums exceed the static scratch of %d: use tv_adj_bernoulli_ll_fwd with a "
                 "workspace", (long long)need, TV_ADJ_STATIC_MAX_PARTIALS);
    return TV_ERR_WORKSPACE;
  }
  float *partial = nullptr;
  rc = adj_static_scratch(&partial);
  if (rc != TV_OK) return rc;
  return adj_fwd_launch(n_samples, batch, max_node_count, dim, h, adj, biased, alpha,
                        log_threshold, partial, ll, (hipStream_t)stream);
}

// how many workgroups share the column blocks of one row block in the backward pass: enough
// to put ~4 workgroups on every CU when there are few (sample, trace) pairs of many nodes
static inline int adj_bwd_jsplit(int64_t KB, int nb) {
  const int most = (nb + ADJ_WAVES - 1) / ADJ_WAVES;          // at least one block per wave
  int js = 1;
  while (js < most && KB * nb * js < 1024) ++js;
  return js;
}

extern "C" int64_t tv_adj_bernoulli_ll_bwd_workspace_bytes(int64_t n_samples, int64_t batch,
                                                           int64_t max_node_count, int64_t dim) {
  if (n_samples < 1 || batch < 1 || max_node_count < 2 || dim < 1) return 0;
  const int js = adj_bwd_jsplit(n_samples * batch, (int)tv_cdiv(max_node_count, AT));
  return js > 1 ? (int64_t)sizeof(float) * n_samples * batch * js * max_node_count * dim : 0;
}

extern "C" int tv_adj_bernoulli_ll_bwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                       int64_t dim, const float *h, const float *adj, int biased,
                                       float alpha, float log_threshold, const float *dll,
                                       float *dh, void *workspace, int64_t workspace_bytes,
                                       void *stream) {
  const int rc = adj_check("tv_adj_bernoulli_ll_bwd", n_samples, batch, max_node_count, dim);
  if (rc != TV_OK) return rc;
  if (batch == 0) return TV_OK;
  TV_REQUIRE(h && adj && dll && dh, "tv_adj_bernoulli_ll_bwd: null pointer");
  int nb, ntiles, ngroups;
  adj_geometry(max_node_count, &nb, &ntiles, &ngroups);
  const int M = (int)max_node_count, D = (int)dim;
  const int64_t KB = n_samples * batch;
  const int js = adj_bwd_jsplit(KB, nb);
  const int64_t need = tv_adj_bernoulli_ll_bwd_workspace_bytes(n_samples, batch, max_node_count, dim);
  if (need > 0 && (!workspace || workspace_bytes < need)) {
    tv_set_error("tv_adj_bernoulli_ll_bwd: workspace of %lld bytes needed, got %lld",
                 (long long)need, (long long)workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  TV_REQUIRE(KB * nb * js < (int64_t)1 << 31, "tv_adj_bernoulli_ll_bwd: grid too large");
  float *out = js > 1 ? (float *)workspace : dh;
  const int nd = (D + AT - 1) / AT;
  const size_t ldsb = sizeof(float) * (ADJ_WAVES * AT * ATS + (ADJ_WAVES - 1) * nd * 16 * 64);
  const dim3 grid((unsigned)(KB * nb * js)), block(64 * ADJ_WAVES);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = dim % 8 == 0;
#define ADJ_BWD(NV, VEC, ND)                                                                   \
  hipLaunchKernelGGL((adj_ll_bwd_kernel<NV, VEC, ND>), grid, block, ldsb, s, batch, M, D, nb,  \
                     js, h, adj, biased, alpha, log_threshold, dll, out)
  if (nd == 1) { if (vec) ADJ_BWD(5, true, 1); else ADJ_BWD(5, false, 1); }
  else if (dim <= 40) { if (vec) ADJ_BWD(5, true, 2); else ADJ_BWD(5, false, 2); }
  else if (nd == 2) { if (vec) ADJ_BWD(10, true, 2); else ADJ_BWD(10, false, 2); }
  else { if (vec) ADJ_BWD(10, true, 3); else ADJ_BWD(10, false, 3); }
#undef ADJ_BWD
  TV_LAUNCH_CHECK("tv_adj_bernoulli_ll_bwd");
  if (js > 1) {
    const int64_t MD = (int64_t)M * D;
    hipLaunchKernelGGL(adj_dh_reduce_kernel, dim3((unsigned)tv_cdiv(KB * MD, 256)), dim3(256), 0,
                       s, KB, MD, js, out, dh);
    TV_LAUNCH_CHECK("tv_adj_bernoulli_ll_bwd (reduce)");
  }
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
                                     const float *dll, float *dlogits, int64_t dlogits_stride,
                                     void *stream) {
  TV_REQUIRE(rows >= 0 && classes >= 1 && target_rows >= 1, "tv_categorical_ll_bwd: bad sizes");
  TV_REQUIRE(dlogits_stride >= classes && dlogits_stride - classes < 64,
             "tv_categorical_ll_bwd: dlogits_stride %lld must be classes .. classes + 63",
             (long long)dlogits_stride);
  if (rows == 0) return TV_OK;
  TV_REQUIRE(rows % target_rows == 0, "tv_categorical_ll_bwd: rows must be a multiple of target_rows");
  TV_REQUIRE(logits && target && lse && dll && dlogits, "tv_categorical_ll_bwd: null pointer");
  hipLaunchKernelGGL(categorical_ll_bwd_kernel, dim3(wave_row_blocks(rows)), dim3(256), 0,
                     (hipStream_t)stream, rows, classes, target_rows, logits, target, biased, alpha,
                     log_threshold, lse, dll, dlogits, dlogits_stride);
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
