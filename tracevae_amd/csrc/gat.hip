// (b) GAT message passing over the destination-sorted CSC: attention scores, edge softmax,
// weighted neighbour sum (forward) and the matching gradients (backward).
//
// Mapping: one 64-lane wave owns one destination row.  A row of H*F floats (1000 by
// default) is held as float4 chunks, chunk q = lane + 64*c, so every neighbour row is one
// fully coalesced 16-B-per-lane gather; a trace's rows sit next to each other, so the
// neighbour rows are L2 hits after the first touch.  Edge scores live one edge per lane
// and are reduced with wave shuffles (segment max / sum).  No float atomics: column
// statistics / parameter gradients are reduced wave -> workgroup (LDS, wave order) ->
// partial slot per workgroup -> fixed-order second stage.
//
// HBM-bound.  Algorithmic bytes, forward, per destination row with d in-edges:
//   4*HF*d (ft gathers) + 4*HF (out) + 8*H*(d+1) (scores) + 8*d (indices) + 4*H*d (alpha).
#include <cstdlib>
#include <type_traits>

#include "tv_common.h"
#include "tv_slots.h"

namespace {

constexpr int MAXH = 4;
#ifndef TV_GAT_WAVES_PER_SIMD
#define TV_GAT_WAVES_PER_SIMD 4
#endif
// rows of up to 1024 floats (CH = 4 float4 per lane) fit 128 registers; the wider form (CH = 8,
// up to 2048 floats: the doubled widths of BASELINE configs[4]) spilled 300-540 bytes per lane
// under that cap and ran at a third of the narrow form's memory rate, so it gets 256
constexpr int gat_waves_per_simd(int ch) { return ch > 4 ? 2 : TV_GAT_WAVES_PER_SIMD; }

// The rows a wave walks.  Workgroups go to the eight XCDs round-robin (id mod 8), and a row's
// neighbours sit a few rows away (a trace's spans are contiguous), so with rows dealt out
// wave by wave over the whole grid a gathered row was fetched into 1.4-1.5 L2s on average
// (PMC: 281 MB read by the forward pass for 205 MB of ft).  Each XCD therefore owns one
// contiguous eighth of the rows and its workgroups stride over that eighth only.
struct RowWalk { int64_t beg, end, step; };
__device__ __forceinline__ RowWalk row_walk(int64_t rows, int nw) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t g = gridDim.x, b = blockIdx.x;
  RowWalk k;
  if ((g & 7) != 0) {                                // few rows: the plain deal
    k.beg = b * nw + w; k.end = rows; k.step = g * nw;
    return k;
  }
  const int64_t per = (rows + 7) >> 3;
  const int64_t lo = (b & 7) * per, hi = lo + per < rows ? lo + per : rows;
  k.beg = lo + (b >> 3) * nw + w; k.end = hi; k.step = (g >> 3) * nw;
  return k;
}
constexpr float NEG_SLOPE = 0.2f;  // GATConv negative_slope (appendix A1)

__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 y) {
  return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), fmaf(a, x.w, y.w));
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
// value of head h out of HT per-head values.  The kernels are specialised on the head count
// (1, 2, or up to MAXH) so that the common cases are a plain register select: a select chain
// over a 4-entry array gets turned into an indexed scratch load by the compiler.
template <int HT>
__device__ __forceinline__ float sel_head(const float (&a)[HT], int h) {
  if (HT == 1) return a[0];
  if (HT == 2) return h ? a[1] : a[0];
  float r = a[0];
#pragma unroll
  for (int k = 1; k < HT; ++k) r = (h == k) ? a[k] : r;
  return r;
}

// Combine S per-wave column vectors (float4 chunks in registers) over the waves of the
// workgroup in wave order and write them to out[S][C].  lds holds S*C floats.
template <int S, int CH>
__device__ __forceinline__ void block_col_reduce(float *lds, const float4 (&v)[S][CH], int c4,
                                                 float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float4 *l4 = reinterpret_cast<float4 *>(lds);
  for (int w = 0; w < nw; ++w) {
    if (wave == w) {
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          if (q < c4) l4[s * c4 + q] = (w == 0) ? v[s][c] : f4_add(l4[s * c4 + q], v[s][c]);
        }
    }
    __syncthreads();
  }
  float4 *o4 = reinterpret_cast<float4 *>(out);
  for (int i = threadIdx.x; i < S * c4; i += blockDim.x) o4[i] = l4[i];
}

// ------------------------------------------------------------------------------------
// el / er
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gat_scores_kernel(int64_t rows, int H, int F,
                                                         const float *__restrict__ ft,
                                                         const float *__restrict__ al,
                                                         const float *__restrict__ ar,
                                                         float *__restrict__ el,
                                                         float *__restrict__ er) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int f4 = F >> 2;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float4 *row = reinterpret_cast<const float4 *>(ft + r * (int64_t)H * F);
    for (int h = 0; h < H; ++h) {
      float sl = 0.f, sr = 0.f;
      for (int q = lane; q < f4; q += 64) {
        const float4 x = row[h * f4 + q];
        sl += f4_dot(x, reinterpret_cast<const float4 *>(al)[h * f4 + q]);
        sr += f4_dot(x, reinterpret_cast<const float4 *>(ar)[h * f4 + q]);
      }
      sl = tv_wave_sum(sl);
      sr = tv_wave_sum(sr);
      if (lane == 0) {
        el[r * H + h] = sl;
        er[r * H + h] = sr;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// forward aggregate
// ------------------------------------------------------------------------------------
struct AggArgs {
  int64_t rows_per_rep, n_rep, rows, num_edges;
  int H, F, c4;  // c4 = H*F/4
  const int64_t *indptr, *indices, *rev;
  const float *ft, *el, *er, *bias, *attn_l, *attn_r, *alpha_in, *dout;
  float *out, *alpha, *stats_partial;
  int64_t stats_slots;       // workgroup slots the caller sized stats_partial for
  const float *row_weight;   // optional weights of the rows in the column statistics
  const float *ep_scale, *ep_shift;
  float ep_slope;
  // backward
  float *dpre, *der, *dft, *partial;
  // optional: 256 partial maxima of |out| (forward with the eval epilogue) or |dft row| (backward
  // with score-gradient columns), kept for the fp16 GEMM forms that read those rows next
  // (tv_gemm_f32_ex).  Workgroups fold their maximum in with an integer atomic max on the bits
  // of the non-negative float: order-free, so the result is the same on every run.
  float *out_max;
  // row strides in floats: ft (H*F dense; H*F + 2H when el / er are two more column groups of the
  // fc product, tv_gat_fc_score_weight), el / er (H dense), dft
  int64_t ld_ft, ld_e, ld_dft;
};

template <int CH, int HT, bool STATS, bool EPILOGUE>
__global__ __launch_bounds__(512, gat_waves_per_simd(CH)) void gat_aggregate_fwd_kernel(AggArgs a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const RowWalk rw = row_walk(a.rows, nw);
  const int H = a.H, F = a.F, c4 = a.c4;
  const int64_t HF = (int64_t)H * F;

  int hc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) hc[c] = ((lane + 64 * c) * 4) / F;

  float4 ssum[STATS ? CH : 1], ssq[STATS ? CH : 1];
  if (STATS) {
#pragma unroll
    for (int c = 0; c < CH; ++c) ssum[c] = ssq[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float omax = 0.f;

  // A row costs a chain of dependent loads: indptr -> indices -> scores -> neighbour rows.  The
  // first two links are taken off the chain by looking ahead: (p0, deg) of the row after next
  // and the neighbour index of the next row are loaded while this row is being gathered.
  int64_t p0_b = 0, s_b = 0, p0_a = 0;
  int deg_b = 0, deg_a = 0;
  if (rw.beg < rw.end) {
    const int64_t v = rw.beg % a.rows_per_rep;
    p0_b = a.indptr[v];
    deg_b = (int)(a.indptr[v + 1] - p0_b);
    s_b = (lane < deg_b) ? a.indices[p0_b + lane] : 0;
  }
  if (rw.beg + rw.step < rw.end) {
    const int64_t v = (rw.beg + rw.step) % a.rows_per_rep;
    p0_a = a.indptr[v];
    deg_a = (int)(a.indptr[v + 1] - p0_a);
  }
  for (int64_t r = rw.beg; r < rw.end; r += rw.step) {
    const int64_t rep = r / a.rows_per_rep;
    const int64_t v = r - rep * a.rows_per_rep;
    const int64_t base = rep * a.rows_per_rep;
    const int64_t p0 = p0_b;
    const int deg = deg_b;
    const int64_t s_rel = s_b;
    p0_b = p0_a;
    deg_b = deg_a;
    if (r + rw.step < rw.end) s_b = (lane < deg_b) ? a.indices[p0_b + lane] : 0;
    if (r + 2 * rw.step < rw.end) {
      const int64_t v2 = (r + 2 * rw.step) % a.rows_per_rep;
      p0_a = a.indptr[v2];
      deg_a = (int)(a.indptr[v2 + 1] - p0_a);
    }
    float erv[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) erv[h] = (h < H) ? a.er[r * a.ld_e + h] : 0.f;

    float4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (deg <= 64) {
      // the usual case (a span has few neighbours): one edge per lane, the index and the score
      // are loaded once and the whole softmax stays in registers
      const bool live = lane < deg;
      const int64_t s = live ? base + s_rel : 0;
      // the first neighbour's row does not depend on the softmax: fetch it under the scores
      float4 first[CH];
      {
        const float4 *srow0 = reinterpret_cast<const float4 *>(a.ft + __shfl(s, 0, 64) * a.ld_ft);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          first[c] = (q < c4) ? srow0[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      float e[HT], al[HT];
#pragma unroll
      for (int h = 0; h < HT; ++h)
        e[h] = (live && h < H) ? tv_lrelu(a.el[s * a.ld_e + h] + erv[h], NEG_SLOPE) : -INFINITY;
#pragma unroll
      for (int h = 0; h < HT; ++h) {
        const float mx = tv_wave_max(e[h]);
        const float ex = live ? expf(e[h] - mx) : 0.f;
        const float den = tv_wave_sum(ex);
        al[h] = ex / den;
        if (a.alpha && live && h < H) a.alpha[(rep * a.num_edges + p0 + lane) * H + h] = al[h];
      }
      if (deg > 0) {
        float aj[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) aj[h] = __shfl(al[h], 0, 64);
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = f4_fma(sel_head<HT>(aj, hc[c]), first[c], acc[c]);
      }
      // two neighbour rows in flight per trip
      int jj = 1;
      for (; jj + 1 < deg; jj += 2) {
        const int64_t sj0 = __shfl(s, jj, 64), sj1 = __shfl(s, jj + 1, 64);
        float aj0[HT], aj1[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) {
          aj0[h] = __shfl(al[h], jj, 64);
          aj1[h] = __shfl(al[h], jj + 1, 64);
        }
        const float4 *srow0 = reinterpret_cast<const float4 *>(a.ft + sj0 * a.ld_ft);
        const float4 *srow1 = reinterpret_cast<const float4 *>(a.ft + sj1 * a.ld_ft);
        float4 x0[CH], x1[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          x0[c] = (q < c4) ? srow0[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          x1[c] = (q < c4) ? srow1[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          acc[c] = f4_fma(sel_head<HT>(aj0, hc[c]), x0[c], acc[c]);
          acc[c] = f4_fma(sel_head<HT>(aj1, hc[c]), x1[c], acc[c]);
        }
      }
      if (jj < deg) {
        const int64_t sj = __shfl(s, jj, 64);
        float aj[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) aj[h] = __shfl(al[h], jj, 64);
        const float4 *srow = reinterpret_cast<const float4 *>(a.ft + sj * a.ld_ft);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          if (q < c4) acc[c] = f4_fma(sel_head<HT>(aj, hc[c]), srow[q], acc[c]);
        }
      }
    } else {
    // pass 1: segment max of the edge scores
    float mx[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) mx[h] = -INFINITY;
    for (int c0 = 0; c0 < deg; c0 += 64) {
      const int j = c0 + lane;
      if (j < deg) {
        const int64_t s = base + a.indices[p0 + j];
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) mx[h] = fmaxf(mx[h], tv_lrelu(a.el[s * a.ld_e + h] + erv[h], NEG_SLOPE));
      }
    }
#pragma unroll
    for (int h = 0; h < HT; ++h) mx[h] = tv_wave_max(mx[h]);
    // pass 2: segment sum of exp
    float den[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) den[h] = 0.f;
    for (int c0 = 0; c0 < deg; c0 += 64) {
      const int j = c0 + lane;
      if (j < deg) {
        const int64_t s = base + a.indices[p0 + j];
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) den[h] += expf(tv_lrelu(a.el[s * a.ld_e + h] + erv[h], NEG_SLOPE) - mx[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < HT; ++h) den[h] = tv_wave_sum(den[h]);

    // pass 3: alpha per edge (one edge per lane), then the weighted neighbour sum
    for (int c0 = 0; c0 < deg; c0 += 64) {
      const int j = c0 + lane;
      int64_t s = 0;
      float al[HT];
#pragma unroll
      for (int h = 0; h < HT; ++h) al[h] = 0.f;
      if (j < deg) {
        s = base + a.indices[p0 + j];
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) {
            al[h] = expf(tv_lrelu(a.el[s * a.ld_e + h] + erv[h], NEG_SLOPE) - mx[h]) / den[h];
            if (a.alpha) a.alpha[(rep * a.num_edges + p0 + j) * H + h] = al[h];
          }
      }
      const int cnt = (deg - c0 < 64) ? (deg - c0) : 64;
      for (int jj = 0; jj < cnt; ++jj) {
        const int64_t sj = __shfl(s, jj, 64);
        float aj[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) aj[h] = __shfl(al[h], jj, 64);
        const float4 *srow = reinterpret_cast<const float4 *>(a.ft + sj * a.ld_ft);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          if (q < c4) acc[c] = f4_fma(sel_head<HT>(aj, hc[c]), srow[q], acc[c]);
        }
      }
    }

    }

    // epilogue: + bias, optional fused BatchNorm(eval)+LeakyReLU or column statistics
    float4 *orow = reinterpret_cast<float4 *>(a.out + r * HF);
    const float sw = (STATS && a.row_weight) ? a.row_weight[v] : 1.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int q = lane + 64 * c;
      if (q < c4) {
        float4 o = f4_add(acc[c], reinterpret_cast<const float4 *>(a.bias)[q]);
        if (STATS) {
          const float4 wo = make_float4(sw * o.x, sw * o.y, sw * o.z, sw * o.w);
          ssum[c] = f4_add(ssum[c], wo);
          ssq[c] = make_float4(fmaf(wo.x, o.x, ssq[c].x), fmaf(wo.y, o.y, ssq[c].y),
                               fmaf(wo.z, o.z, ssq[c].z), fmaf(wo.w, o.w, ssq[c].w));
        }
        if (EPILOGUE) {
          const float4 sc = reinterpret_cast<const float4 *>(a.ep_scale)[q];
          const float4 sh = reinterpret_cast<const float4 *>(a.ep_shift)[q];
          o.x = tv_lrelu(fmaf(o.x, sc.x, sh.x), a.ep_slope);
          o.y = tv_lrelu(fmaf(o.y, sc.y, sh.y), a.ep_slope);
          o.z = tv_lrelu(fmaf(o.z, sc.z, sh.z), a.ep_slope);
          o.w = tv_lrelu(fmaf(o.w, sc.w, sh.w), a.ep_slope);
          omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        // many replicas (K-sample scoring: gigabytes of output that no cache will hold for the
        // next kernel): stream the rows past the L2, which then keeps the gathered rows
        if (EPILOGUE && a.n_rep >= 8) {
          typedef float f32x4_t __attribute__((ext_vector_type(4)));
          const f32x4_t ov = {o.x, o.y, o.z, o.w};
          __builtin_nontemporal_store(ov, reinterpret_cast<f32x4_t *>(orow + q));
        } else {
          orow[q] = o;
        }
      }
    }
  }

  if (STATS) {
    float4 v[2][CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { v[0][c] = ssum[c]; v[1][c] = ssq[c]; }
    block_col_reduce<2, CH>(lds, v, c4, a.stats_partial + (int64_t)blockIdx.x * 2 * HF);
  }
  if (EPILOGUE && a.out_max) {
    omax = tv_wave_max(omax);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(a.out_max) + (blockIdx.x & 255), __float_as_uint(omax));
  }
}

// ------------------------------------------------------------------------------------
// forward aggregate, windowed: a workgroup stages LDS_ROWS consecutive [ft | el | er] rows in LDS
// and serves every gather whose source lies in the window from there.
//
// Why: the counters say the per-row kernel above is bound by the read requests a CU can keep in
// flight (TCP -> L2: 40-70 outstanding per CU in every memory-bound kernel of this library,
// whatever its structure; latency 660 cycles here), and a row with d in-edges sends d rows
// through that queue although the L2 serves most of them.  A trace's rows are contiguous and
// 65 % of its edges join rows less than a 32-row window apart, so the window cuts the rows
// through the L1 from 2.0 to 1.34 per output row on the decoder graph and from 2.96 to 1.69 on
// the encoder graph.  The next window's rows are fetched into registers while this window is
// summed (a plain streaming read: every load of a wave in flight together).
// Measured (tools/gat_lab.py, 1.28 M rows of K = 100 scoring): 2800 -> 2500 us per launch, K = 100
// scoring 73.0 -> 71.5 ms on the same box; bit-equal to the per-row kernel (same edge order).
// Taken for the column layout of tv_gat_fc_score_weight ([ft | el | er] rows), H*F + 2H <= 1024,
// no column statistics (the training variant needs 32 more registers per lane and spills in
// this form) and at least 512 windows (TV_GAT_LDS_MIN_WINDOWS in the environment moves that
// threshold, for A/B runs); everything else stays on the per-row kernel.
// ------------------------------------------------------------------------------------
#ifndef TV_GAT_LDS_ROWS
#define TV_GAT_LDS_ROWS 32
#endif
constexpr int LDS_ROWS = TV_GAT_LDS_ROWS;
constexpr int LDS_WAVES = LDS_ROWS / 2;        // two rows of the window per wave
constexpr int LDS_WGS_PER_CU = 32 / LDS_ROWS;  // by LDS: 32 rows of 4 KB fill a CU

template <int HT, bool STATS, bool EPILOGUE>
__global__ __launch_bounds__(64 * LDS_WAVES, 4) void gat_aggregate_fwd_lds_kernel(AggArgs a) {
  extern __shared__ float lds[];
  constexpr int CH = 4;
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int H = a.H, F = a.F, c4 = a.c4;
  const int64_t HF = (int64_t)H * F;
  const int ld4 = (int)(a.ld_ft >> 2);                    // float4 per staged row (<= 256)
  const int ldf = (int)a.ld_ft, hf = (int)HF;             // (LDS is indexed with 32-bit integers)
  float4 *win = reinterpret_cast<float4 *>(lds);          // [LDS_ROWS][ld4]
  float *red = lds + LDS_ROWS * ldf;          // column-statistics scratch (STATS)

  int hc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) hc[c] = ((lane + 64 * c) * 4) / F;
  float4 ssum[STATS ? CH : 1], ssq[STATS ? CH : 1];
  if (STATS) {
#pragma unroll
    for (int c = 0; c < CH; ++c) ssum[c] = ssq[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float omax = 0.f;

  // windows: each XCD owns a contiguous eighth of them (see row_walk)
  const int64_t nwin = (a.rows + LDS_ROWS - 1) / LDS_ROWS;
  int64_t wbeg, wend, wstep;
  {
    const int64_t g = gridDim.x, b = blockIdx.x;
    if ((g & 7) != 0) { wbeg = b; wend = nwin; wstep = g; }
    else {
      const int64_t per = (nwin + 7) >> 3;
      const int64_t lo = (b & 7) * per;
      wbeg = lo + (b >> 3); wend = lo + per < nwin ? lo + per : nwin; wstep = g >> 3;
    }
  }
  // the rows of the NEXT window, on their way: two rows per wave
  float4 pf[2][CH];
  auto fetch = [&](int64_t wi) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t r = wi * LDS_ROWS + w + LDS_WAVES * j;
      const float4 *src = reinterpret_cast<const float4 *>(a.ft + (r < a.rows ? r : 0) * a.ld_ft);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = lane + 64 * c;
        pf[j][c] = (q < ld4 && r < a.rows) ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  // the index chain of a row (indptr -> indices -> el) is looked ahead like the rows: (p0, deg)
  // of the rows two windows on and the neighbour indices of the next window's rows are loaded
  // while this window is summed, and the el loads of both rows of a wave go out together
  // (32-bit: a replica has fewer than 2^31 rows and edges, checked on the host)
  int p0_c[2] = {0, 0}, p0_1[2] = {0, 0}, p0_2[2] = {0, 0}, s_c[2] = {0, 0}, s_1[2] = {0, 0};
  int deg_c[2] = {0, 0}, deg_1[2] = {0, 0}, deg_2[2] = {0, 0};
  auto load_pd = [&](int64_t wi, int j, int &p0, int &deg) {
    // (no branch: the loads of a window are a fixed sequence, so that the wait for the FIRST of
    // them - the el values the sums start with - can leave every later one in flight; on gfx9 a
    // wave's loads return in order and a wait is a count of the younger ones)
    const int64_t r = wi * LDS_ROWS + w + LDS_WAVES * j;
    const int64_t v = (r < a.rows ? r : a.rows - 1) % a.rows_per_rep;
    p0 = (int)a.indptr[v];
    deg = (r < a.rows) ? (int)a.indptr[v + 1] - p0 : 0;
  };
  if (wbeg < wend) {
    fetch(wbeg);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      load_pd(wbeg, j, p0_c[j], deg_c[j]);
      s_c[j] = (lane < deg_c[j]) ? (int)a.indices[p0_c[j] + lane] : 0;
      if (wbeg + wstep < wend) load_pd(wbeg + wstep, j, p0_1[j], deg_1[j]);
    }
  }
  for (int64_t wi = wbeg; wi < wend; wi += wstep) {
    const int64_t r0 = wi * LDS_ROWS;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = lane + 64 * c;
        if (q < ld4) win[(w + LDS_WAVES * j) * ld4 + q] = pf[j][c];
      }
    __syncthreads();
    // el of every edge of the wave's two rows first (one edge per lane; rows of more than 64
    // neighbours read theirs again below), then the look-ahead loads behind them
    float elv[2][HT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t r = r0 + w + LDS_WAVES * j;
      const int64_t base = (r < a.rows ? r / a.rows_per_rep : 0) * a.rows_per_rep;
      const int64_t srow = (lane < deg_c[j] && deg_c[j] <= 64) ? base + s_c[j] : 0;
#pragma unroll
      for (int h = 0; h < HT; ++h) elv[j][h] = a.ft[srow * a.ld_ft + HF + (h < H ? h : 0)];
    }
    {
      // (past the last window the same loads go to the last window again and are not used)
      const int64_t w1 = wi + wstep < wend ? wi + wstep : wi;
      const int64_t w2 = wi + 2 * wstep < wend ? wi + 2 * wstep : wi;
      fetch(w1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lim = wi + wstep < wend ? deg_1[j] : 0;
        s_1[j] = (lane < lim) ? (int)a.indices[p0_1[j] + lane] : 0;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) load_pd(w2, j, p0_2[j], deg_2[j]);
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lr = w + LDS_WAVES * j;                    // row of the window
      const int64_t r = r0 + lr;
      if (r < a.rows) {
      const int64_t rep = r / a.rows_per_rep;
      const int64_t v = r - rep * a.rows_per_rep;
      const int64_t base = rep * a.rows_per_rep;
      const int64_t p0 = p0_c[j];
      const int deg = deg_c[j];
      const float *own = lds + lr * ldf;
      float erv[HT];
#pragma unroll
      for (int h = 0; h < HT; ++h) erv[h] = (h < H) ? own[hf + H + h] : 0.f;
      float4 acc[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      // el of a source row: one float per head and edge, read from memory (an L2 hit; a per-lane
      // choice between an LDS and a global load does not get through this compiler)
      auto el_of = [&](int64_t s, int h) -> float { return a.ft[s * a.ld_ft + HF + h]; };
      auto add_row = [&](int64_t sj, const float (&aj)[HT]) {
        const int64_t d = sj - r0;                         // wave-uniform
        if (d >= 0 && d < LDS_ROWS) {
          const float4 *srow = win + (int)d * ld4;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int q = lane + 64 * c;
            if (q < c4) acc[c] = f4_fma(sel_head<HT>(aj, hc[c]), srow[q], acc[c]);
          }
        } else {
          const float4 *srow = reinterpret_cast<const float4 *>(a.ft + sj * a.ld_ft);
          float4 x[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int q = lane + 64 * c;
            x[c] = (q < c4) ? srow[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[c] = f4_fma(sel_head<HT>(aj, hc[c]), x[c], acc[c]);
        }
      };
      if (deg <= 64) {
        const bool live = lane < deg;
        const int64_t s = live ? base + s_c[j] : 0;
        float e[HT], al[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h)
          e[h] = (live && h < H) ? tv_lrelu(elv[j][h] + erv[h], NEG_SLOPE) : -INFINITY;
#pragma unroll
        for (int h = 0; h < HT; ++h) {
          const float mx = tv_wave_max(e[h]);
          const float ex = live ? expf(e[h] - mx) : 0.f;
          const float den = tv_wave_sum(ex);
          al[h] = ex / den;
          if (a.alpha && live && h < H) a.alpha[(rep * a.num_edges + p0 + lane) * H + h] = al[h];
        }
        for (int jj = 0; jj < deg; ++jj) {                 // summed in edge order
          const int64_t sj = __shfl(s, jj, 64);
          float aj[HT];
#pragma unroll
          for (int h = 0; h < HT; ++h) aj[h] = __shfl(al[h], jj, 64);
          add_row(sj, aj);
        }
      } else {
        // many neighbours (rare): three passes over the edge list, 64 edges at a time
        float mx[HT], den[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) { mx[h] = -INFINITY; den[h] = 0.f; }
        for (int c0 = 0; c0 < deg; c0 += 64) {
          const int jx = c0 + lane;
          if (jx < deg) {
            const int64_t s = base + a.indices[p0 + jx];
#pragma unroll
            for (int h = 0; h < HT; ++h)
              if (h < H) mx[h] = fmaxf(mx[h], tv_lrelu(el_of(s, h) + erv[h], NEG_SLOPE));
          }
        }
#pragma unroll
        for (int h = 0; h < HT; ++h) mx[h] = tv_wave_max(mx[h]);
        for (int c0 = 0; c0 < deg; c0 += 64) {
          const int jx = c0 + lane;
          if (jx < deg) {
            const int64_t s = base + a.indices[p0 + jx];
#pragma unroll
            for (int h = 0; h < HT; ++h)
              if (h < H) den[h] += expf(tv_lrelu(el_of(s, h) + erv[h], NEG_SLOPE) - mx[h]);
          }
        }
#pragma unroll
        for (int h = 0; h < HT; ++h) den[h] = tv_wave_sum(den[h]);
        for (int c0 = 0; c0 < deg; c0 += 64) {
          const int jx = c0 + lane;
          int64_t s = 0;
          float al[HT];
#pragma unroll
          for (int h = 0; h < HT; ++h) al[h] = 0.f;
          if (jx < deg) {
            s = base + a.indices[p0 + jx];
#pragma unroll
            for (int h = 0; h < HT; ++h)
              if (h < H) {
                al[h] = expf(tv_lrelu(el_of(s, h) + erv[h], NEG_SLOPE) - mx[h]) / den[h];
                if (a.alpha) a.alpha[(rep * a.num_edges + p0 + jx) * H + h] = al[h];
              }
          }
          const int cnt = (deg - c0 < 64) ? (deg - c0) : 64;
          for (int jj = 0; jj < cnt; ++jj) {
            const int64_t sj = __shfl(s, jj, 64);
            float aj[HT];
#pragma unroll
            for (int h = 0; h < HT; ++h) aj[h] = __shfl(al[h], jj, 64);
            add_row(sj, aj);
          }
        }
      }
      // epilogue: + bias, optional fused BatchNorm(eval)+LeakyReLU or column statistics
      float4 *orow = reinterpret_cast<float4 *>(a.out + r * HF);
      const float sw = (STATS && a.row_weight) ? a.row_weight[v] : 1.0f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = lane + 64 * c;
        if (q < c4) {
          float4 o = f4_add(acc[c], reinterpret_cast<const float4 *>(a.bias)[q]);
          if (STATS) {
            const float4 wo = make_float4(sw * o.x, sw * o.y, sw * o.z, sw * o.w);
            ssum[c] = f4_add(ssum[c], wo);
            ssq[c] = make_float4(fmaf(wo.x, o.x, ssq[c].x), fmaf(wo.y, o.y, ssq[c].y),
                                 fmaf(wo.z, o.z, ssq[c].z), fmaf(wo.w, o.w, ssq[c].w));
          }
          if (EPILOGUE) {
            const float4 sc = reinterpret_cast<const float4 *>(a.ep_scale)[q];
            const float4 sh = reinterpret_cast<const float4 *>(a.ep_shift)[q];
            o.x = tv_lrelu(fmaf(o.x, sc.x, sh.x), a.ep_slope);
            o.y = tv_lrelu(fmaf(o.y, sc.y, sh.y), a.ep_slope);
            o.z = tv_lrelu(fmaf(o.z, sc.z, sh.z), a.ep_slope);
            o.w = tv_lrelu(fmaf(o.w, sc.w, sh.w), a.ep_slope);
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
          }
          if (EPILOGUE && a.n_rep >= 8) {
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            const f32x4_t ov = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(ov, reinterpret_cast<f32x4_t *>(orow + q));
          } else {
            orow[q] = o;
          }
        }
      }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      p0_c[j] = p0_1[j]; deg_c[j] = deg_1[j]; s_c[j] = s_1[j];
      p0_1[j] = p0_2[j]; deg_1[j] = deg_2[j];
    }
    __syncthreads();                                       // the window is overwritten next
  }

  if (STATS) {
    float4 v[2][CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { v[0][c] = ssum[c]; v[1][c] = ssq[c]; }
    block_col_reduce<2, CH>(red, v, c4, a.stats_partial + (int64_t)blockIdx.x * 2 * HF);
    // the caller sized the partials for stats_slots workgroups: the slots behind this grid are zero
    for (int64_t sl = (int64_t)gridDim.x + blockIdx.x; sl < a.stats_slots; sl += gridDim.x)
      for (int64_t i = threadIdx.x; i < 2 * HF; i += blockDim.x) a.stats_partial[sl * 2 * HF + i] = 0.f;
  }
  if (EPILOGUE && a.out_max) {
    omax = tv_wave_max(omax);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(a.out_max) + (blockIdx.x & 255), __float_as_uint(omax));
  }
}

// ------------------------------------------------------------------------------------
// backward, step A (per destination row): gradient of the edge scores
//   g_j   = <dout[v,h,:], ft[u_j,h,:]>
//   de_j  = alpha_j * (g_j - sum_k alpha_k g_k)            (softmax)
//   dpre_j= de_j * (pre_j > 0 ? 1 : 0.2)                    (LeakyReLU), pre_j = el[u_j]+er[v]
//   der[v]= sum_j dpre_j ;  dpre kept per CSC position for step B
// ------------------------------------------------------------------------------------
template <int CH, int HT>
__global__ __launch_bounds__(512) void gat_bwd_edge_kernel(AggArgs a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const RowWalk rw = row_walk(a.rows, nw);
  const int H = a.H, F = a.F, c4 = a.c4;
  const int64_t HF = (int64_t)H * F;
  int hc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) hc[c] = ((lane + 64 * c) * 4) / F;

  // dbias = column sums of dout: this kernel holds every dout row once, so the sum rides here
  // (the node pass, which is short of registers, keeps only the two attention gradients)
  float4 pb[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) pb[c] = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int64_t r = rw.beg; r < rw.end; r += rw.step) {
    const int64_t rep = r / a.rows_per_rep;
    const int64_t v = r - rep * a.rows_per_rep;
    const int64_t base = rep * a.rows_per_rep;
    const int64_t p0 = a.indptr[v];
    const int deg = (int)(a.indptr[v + 1] - p0);
    const int64_t ebase = rep * a.num_edges + p0;

    float4 d[CH];
    const float4 *drow = reinterpret_cast<const float4 *>(a.dout + r * HF);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int q = lane + 64 * c;
      d[c] = (q < c4) ? drow[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      pb[c] = f4_add(pb[c], d[c]);
    }

    // pass 1: g_j for every in-edge (kept by lane j; spilled to dpre when deg > 64),
    // and s[h] = sum_j alpha_j g_j
    float sacc[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) sacc[h] = 0.f;
    float gkeep[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) gkeep[h] = 0.f;
    for (int c0 = 0; c0 < deg; c0 += 64) {
      const int cnt = (deg - c0 < 64) ? (deg - c0) : 64;
      const int j = c0 + lane;
      const int64_t s = (j < deg) ? (base + a.indices[p0 + j]) : 0;
      float gj[HT];
#pragma unroll
      for (int h = 0; h < HT; ++h) gj[h] = 0.f;
      // two neighbour rows (and their wave reductions) in flight per trip (four measured slower:
      // the extra registers cost more than the third and fourth load in flight gain)
      auto edges = [&](auto nconst, int jj) {
        constexpr int U = decltype(nconst)::value;
        float4 x[U][CH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float4 *srow = reinterpret_cast<const float4 *>(a.ft + __shfl(s, jj + u, 64) * a.ld_ft);
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int q = lane + 64 * c;
            x[u][c] = (q < c4) ? srow[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
        float part[U][HT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int h = 0; h < HT; ++h) part[u][h] = 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const float dp = f4_dot(d[c], x[u][c]);
#pragma unroll
            for (int h = 0; h < HT; ++h) part[u][h] += (hc[c] == h) ? dp : 0.f;
          }
        }
#pragma unroll
        for (int h = 0; h < HT; ++h) {
          if (h < H) {
            float tot[U];
#pragma unroll
            for (int u = 0; u < U; ++u) tot[u] = tv_wave_sum(part[u][h]);
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (lane == jj + u) gj[h] = tot[u];
          }
        }
      };
      int jj = 0;
      for (; jj + 2 <= cnt; jj += 2) edges(std::integral_constant<int, 2>(), jj);
      if (jj < cnt) edges(std::integral_constant<int, 1>(), jj);
      if (j < deg) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) {
            sacc[h] += a.alpha_in[(ebase + j) * H + h] * gj[h];
            if (deg > 64) a.dpre[(ebase + j) * H + h] = gj[h];
            gkeep[h] = gj[h];
          }
      }
    }
#pragma unroll
    for (int h = 0; h < HT; ++h) sacc[h] = tv_wave_sum(sacc[h]);

    // pass 2: softmax + LeakyReLU backward
    float dsum[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) dsum[h] = 0.f;
    for (int c0 = 0; c0 < deg; c0 += 64) {
      const int j = c0 + lane;
      if (j < deg) {
        const int64_t s = base + a.indices[p0 + j];
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) {
            const float g = (deg > 64) ? a.dpre[(ebase + j) * H + h] : gkeep[h];
            const float al = a.alpha_in[(ebase + j) * H + h];
            const float pre = a.el[s * a.ld_e + h] + a.er[r * a.ld_e + h];
            const float dp = al * (g - sacc[h]) * (pre > 0.f ? 1.f : NEG_SLOPE);
            a.dpre[(ebase + j) * H + h] = dp;
            dsum[h] += dp;
          }
      }
    }
#pragma unroll
    for (int h = 0; h < HT; ++h) {
      if (h < H) {
        const float tot = tv_wave_sum(dsum[h]);
        if (lane == 0) a.der[r * H + h] = tot;
      }
    }
  }

  float4 vb[1][CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) vb[0][c] = pb[c];
  block_col_reduce<1, CH>(lds, vb, c4, a.partial + ((int64_t)blockIdx.x * 3 + 2) * HF);
}

// ------------------------------------------------------------------------------------
// backward, step B (per source row u): the graph is symmetric, so the out-edges of u are
// the opposites of its in-edges: CSC entry p = (w -> u) has the opposite edge (u -> w) at
// CSC position rev[p].
//   dft[u]  = sum_p alpha[rev[p]] * dout[w_p]  +  del[u] * attn_l + der[u] * attn_r
//   del[u]  = sum_p dpre[rev[p]]
//   partial sums for dattn_l (del*ft), dattn_r (der*ft), dbias (dout)
// ------------------------------------------------------------------------------------
// (When el / er are columns of the fc product, [ft | el | er], their gradients leave as columns of
// dft instead and steps A and B run as one pass: gat_bwd_fused_kernel below.)
template <int CH, int HT>
__global__ __launch_bounds__(512, gat_waves_per_simd(CH)) void gat_bwd_node_kernel(AggArgs a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const RowWalk rw = row_walk(a.rows, nw);
  const int H = a.H, F = a.F, c4 = a.c4;
  const int64_t HF = (int64_t)H * F;
  int hc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) hc[c] = ((lane + 64 * c) * 4) / F;

  float4 pal[CH], par[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) pal[c] = par[c] = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int64_t r = rw.beg; r < rw.end; r += rw.step) {
    const int64_t rep = r / a.rows_per_rep;
    const int64_t u = r - rep * a.rows_per_rep;
    const int64_t base = rep * a.rows_per_rep;
    const int64_t p0 = a.indptr[u];
    const int deg = (int)(a.indptr[u + 1] - p0);
    const int64_t eoff = rep * a.num_edges;

    float4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    float del[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) del[h] = 0.f;

    for (int c0 = 0; c0 < deg; c0 += 64) {
      const int j = c0 + lane;
      int64_t w = 0;
      float al[HT], dp[HT];
#pragma unroll
      for (int h = 0; h < HT; ++h) al[h] = dp[h] = 0.f;
      if (j < deg) {
        w = base + a.indices[p0 + j];
        const int64_t rp = eoff + a.rev[p0 + j];
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) {
            al[h] = a.alpha_in[rp * H + h];
            dp[h] = a.dpre[rp * H + h];
          }
      }
#pragma unroll
      for (int h = 0; h < HT; ++h) del[h] += dp[h];
      const int cnt = (deg - c0 < 64) ? (deg - c0) : 64;
      // two neighbour rows in flight per trip
      int jj = 0;
      for (; jj + 1 < cnt; jj += 2) {
        const int64_t wj0 = __shfl(w, jj, 64), wj1 = __shfl(w, jj + 1, 64);
        float aj0[HT], aj1[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) {
          aj0[h] = __shfl(al[h], jj, 64);
          aj1[h] = __shfl(al[h], jj + 1, 64);
        }
        const float4 *wrow0 = reinterpret_cast<const float4 *>(a.dout + wj0 * HF);
        const float4 *wrow1 = reinterpret_cast<const float4 *>(a.dout + wj1 * HF);
        float4 x0[CH], x1[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          x0[c] = (q < c4) ? wrow0[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          x1[c] = (q < c4) ? wrow1[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          acc[c] = f4_fma(sel_head<HT>(aj0, hc[c]), x0[c], acc[c]);
          acc[c] = f4_fma(sel_head<HT>(aj1, hc[c]), x1[c], acc[c]);
        }
      }
      if (jj < cnt) {
        const int64_t wj = __shfl(w, jj, 64);
        float aj[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) aj[h] = __shfl(al[h], jj, 64);
        const float4 *wrow = reinterpret_cast<const float4 *>(a.dout + wj * HF);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = lane + 64 * c;
          if (q < c4) acc[c] = f4_fma(sel_head<HT>(aj, hc[c]), wrow[q], acc[c]);
        }
      }
    }
    float der[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) {
      del[h] = tv_wave_sum(del[h]);
      der[h] = (h < H) ? a.der[r * H + h] : 0.f;
    }

    float4 *orow = reinterpret_cast<float4 *>(a.dft + r * a.ld_dft);
    const float4 *frow = reinterpret_cast<const float4 *>(a.ft + r * a.ld_ft);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int q = lane + 64 * c;
      if (q < c4) {
        const float dl = sel_head<HT>(del, hc[c]), dr = sel_head<HT>(der, hc[c]);
        const float4 wl = reinterpret_cast<const float4 *>(a.attn_l)[q];
        const float4 wr = reinterpret_cast<const float4 *>(a.attn_r)[q];
        orow[q] = f4_fma(dr, wr, f4_fma(dl, wl, acc[c]));
        const float4 f = frow[q];
        pal[c] = f4_fma(dl, f, pal[c]);
        par[c] = f4_fma(dr, f, par[c]);
      }
    }
  }

  float4 v[2][CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { v[0][c] = pal[c]; v[1][c] = par[c]; }
  block_col_reduce<2, CH>(lds, v, c4, a.partial + (int64_t)blockIdx.x * 3 * HF);
}

// ------------------------------------------------------------------------------------
// backward with the score gradients as columns of dft ([dft | del | der | 0], the el / er columns
// of the fc product: tv_gat_fc_score_weight), both steps in ONE pass over
// the rows: the graph is symmetric, so the in-neighbours of u (step A gathers their ft rows for
// the score gradients of u's in-edges) are also its out-neighbours (step B gathers their dout
// rows for dft[u]).  One wave per row u reads dout[u], then ft[s_j] and dout[s_j] of every
// neighbour together: dout is then fetched from memory once (the own-row read and the gathered
// reads of the same rows meet in the L2; u is its own neighbour through the self-loop) where
// the two-kernel form read it in both, 205 MB apart.  What cannot ride along is del[u], the sum
// of dpre over the edges OUT of u: those belong to other rows' softmax and are finished only
// when the pass is; gat_bwd_tail_kernel adds them up afterwards (edge scalars only) and
// writes the [del | der | 0] columns.
// ------------------------------------------------------------------------------------
template <int CH, int HT>
__global__ __launch_bounds__(512) void gat_bwd_fused_kernel(AggArgs a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const RowWalk rw = row_walk(a.rows, nw);
  const int H = a.H, F = a.F, c4 = a.c4;
  const int64_t HF = (int64_t)H * F;
  int hc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) hc[c] = ((lane + 64 * c) * 4) / F;

  float4 pb[CH];                                     // dbias = column sums of dout
#pragma unroll
  for (int c = 0; c < CH; ++c) pb[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  float dmax = 0.f;

  for (int64_t r = rw.beg; r < rw.end; r += rw.step) {  // (a single replica: row = node)
    const int64_t p0 = a.indptr[r];
    const int deg = (int)(a.indptr[r + 1] - p0);

    float4 d[CH], acc[CH];
    const float4 *drow = reinterpret_cast<const float4 *>(a.dout + r * HF);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int q = lane + 64 * c;
      d[c] = (q < c4) ? drow[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      pb[c] = f4_add(pb[c], d[c]);
      acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // A row whose only in-edge is its self-loop (a padded slot of the struct decoder: about half
    // of its rows on the bench's traces, most of them on short real ones): the softmax over one
    // edge has no gradient - alpha = 1, dpre = alpha (g - alpha g) = +0 exactly for any finite g -
    // so ft[r], which only feeds g, is not read: two row transfers instead of three, with the
    // bits of the general path (alpha * dout[r] through the same fma).
    if (deg == 1 && a.indices[p0] == r) {
      float al1[HT];
      const int64_t rp = a.rev[p0];
#pragma unroll
      for (int h = 0; h < HT; ++h) al1[h] = (h < H) ? a.alpha_in[rp * H + h] : 0.f;
      if (lane < H) {
        a.dpre[p0 * H + lane] = 0.f;
        a.der[r * H + lane] = 0.f;
      }
      float4 *orow = reinterpret_cast<float4 *>(a.dft + r * a.ld_dft);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = lane + 64 * c;
        if (q < c4) {
          const float4 o = f4_fma(sel_head<HT>(al1, hc[c]), d[c], acc[c]);
          orow[q] = o;
          dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
      }
      continue;
    }

    float sacc[HT], gkeep[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) sacc[h] = gkeep[h] = 0.f;
    for (int c0 = 0; c0 < deg; c0 += 64) {
      const int cnt = (deg - c0 < 64) ? (deg - c0) : 64;
      const int j = c0 + lane;
      int64_t s = 0;
      float al[HT], gj[HT];
#pragma unroll
      for (int h = 0; h < HT; ++h) al[h] = gj[h] = 0.f;
      if (j < deg) {
        s = a.indices[p0 + j];
        const int64_t rp = a.rev[p0 + j];            // the opposite edge u -> s_j
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) al[h] = a.alpha_in[rp * H + h];
      }
      // U neighbours per trip: their 2U rows (ft and dout) in flight together
      auto edges = [&](auto nconst, int jj) {
        constexpr int U = decltype(nconst)::value;
        float4 x[U][CH], y[U][CH];
        float aj[U][HT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t sj = __shfl(s, jj + u, 64);
#pragma unroll
          for (int h = 0; h < HT; ++h) aj[u][h] = __shfl(al[h], jj + u, 64);
          const float4 *frow = reinterpret_cast<const float4 *>(a.ft + sj * a.ld_ft);
          const float4 *wrow = reinterpret_cast<const float4 *>(a.dout + sj * HF);
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int q = lane + 64 * c;
            x[u][c] = (q < c4) ? frow[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          // the self-loop: dout[u] is already here (every row has one, a padded decoder slot
          // nothing else; the kernel is bound by the read requests a CU keeps in flight, §4f, and
          // this row is one in 1 + 2 deg of them)
          if (sj == r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) y[u][c] = d[c];
          } else {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const int q = lane + 64 * c;
              y[u][c] = (q < c4) ? wrow[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
        float part[U][HT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int h = 0; h < HT; ++h) part[u][h] = 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const float dp = f4_dot(d[c], x[u][c]);
#pragma unroll
            for (int h = 0; h < HT; ++h) part[u][h] += (hc[c] == h) ? dp : 0.f;
            acc[c] = f4_fma(sel_head<HT>(aj[u], hc[c]), y[u][c], acc[c]);
          }
        }
#pragma unroll
        for (int h = 0; h < HT; ++h) {
          if (h < H) {
            float tot[U];
#pragma unroll
            for (int u = 0; u < U; ++u) tot[u] = tv_wave_sum(part[u][h]);
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (lane == jj + u) gj[h] = tot[u];
          }
        }
      };
      // (one neighbour per trip: two per trip, four rows in flight, cost a wave per SIMD in
      // registers and measured 1.3-1.8 % slower on the step, before and after the XCD-aware walk)
      for (int jj = 0; jj < cnt; ++jj) edges(std::integral_constant<int, 1>(), jj);
      if (j < deg) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) {
            sacc[h] += a.alpha_in[(p0 + j) * H + h] * gj[h];
            if (deg > 64) a.dpre[(p0 + j) * H + h] = gj[h];
            gkeep[h] = gj[h];
          }
      }
    }
#pragma unroll
    for (int h = 0; h < HT; ++h) sacc[h] = tv_wave_sum(sacc[h]);

    // softmax + LeakyReLU backward of the in-edges (as gat_bwd_edge_kernel)
    float dsum[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) dsum[h] = 0.f;
    for (int c0 = 0; c0 < deg; c0 += 64) {
      const int j = c0 + lane;
      if (j < deg) {
        const int64_t s = a.indices[p0 + j];
#pragma unroll
        for (int h = 0; h < HT; ++h)
          if (h < H) {
            const float g = (deg > 64) ? a.dpre[(p0 + j) * H + h] : gkeep[h];
            const float alj = a.alpha_in[(p0 + j) * H + h];
            const float pre = a.el[s * a.ld_e + h] + a.er[r * a.ld_e + h];
            const float dp = alj * (g - sacc[h]) * (pre > 0.f ? 1.f : NEG_SLOPE);
            a.dpre[(p0 + j) * H + h] = dp;
            dsum[h] += dp;
          }
      }
    }
#pragma unroll
    for (int h = 0; h < HT; ++h) {
      if (h < H) {
        const float tot = tv_wave_sum(dsum[h]);
        if (lane == 0) a.der[r * H + h] = tot;
      }
    }

    float4 *orow = reinterpret_cast<float4 *>(a.dft + r * a.ld_dft);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int q = lane + 64 * c;
      if (q < c4) {
        orow[q] = acc[c];
        dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(acc[c].x), fabsf(acc[c].y))),
                     fmaxf(fabsf(acc[c].z), fabsf(acc[c].w)));
      }
    }
  }
  if (a.out_max) {
    dmax = tv_wave_max(dmax);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(a.out_max) + (blockIdx.x & 255), __float_as_uint(dmax));
  }

  float4 vb[1][CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) vb[0][c] = pb[c];
  block_col_reduce<1, CH>(lds, vb, c4, a.partial + ((int64_t)blockIdx.x * 3 + 2) * HF);
}

// [del | der | 0] behind the HF gradient columns of every dft row: del[u] = sum of dpre over the
// edges out of u (the opposites, rev[p], of its in-edges), der[u] from the pass above
__device__ __forceinline__ void gat_bwd_score_cols_block(const AggArgs &a, int64_t block) {
  // four lanes per row, each takes every fourth in-edge (a span has few neighbours: one dependent
  // rev -> dpre chain per lane), summed by two shuffles in a fixed order
  const int64_t t = block * blockDim.x + threadIdx.x;
  const int64_t r = t >> 2;
  const int k = (int)(t & 3);
  const bool live = r < a.rows;
  const int H = a.H;
  float del[MAXH];
#pragma unroll
  for (int h = 0; h < MAXH; ++h) del[h] = 0.f;
  if (live) {
    const int64_t p0 = a.indptr[r], p1 = a.indptr[r + 1];
    for (int64_t p = p0 + k; p < p1; p += 4) {
      const int64_t rp = a.rev[p];
#pragma unroll
      for (int h = 0; h < MAXH; ++h)
        if (h < H) del[h] += a.dpre[rp * H + h];
    }
  }
#pragma unroll
  for (int h = 0; h < MAXH; ++h) {
    del[h] += __shfl_xor(del[h], 1, 64);
    del[h] += __shfl_xor(del[h], 2, 64);
  }
  float vmax = 0.f;
  if (live) {
    float *o = a.dft + r * a.ld_dft + (int64_t)H * a.F;
    const int ncol = (2 * H + 3) & ~3;
    for (int c = k; c < ncol; c += 4) {              // lane k writes columns k, k + 4, ...
      float v = 0.f;
#pragma unroll
      for (int h = 0; h < MAXH; ++h) {
        if (c == h && h < H) v = del[h];
        if (c == H + h && h < H) v = a.der[r * H + h];
      }
      o[c] = v;
      vmax = fmaxf(vmax, fabsf(v));
    }
  }
  if (a.out_max) {
    vmax = tv_wave_max(vmax);
    if ((threadIdx.x & 63) == 0)
      atomicMax(reinterpret_cast<unsigned *>(a.out_max) + (block & 255), __float_as_uint(vmax));
  }
}

// out[s*C + c] = sum_b partial[b][s][c] in workgroup order (float64 accumulation)
__device__ __forceinline__ void reduce_partials_block(int64_t block, int64_t nblk, int64_t sc,
                                                      const float *__restrict__ partial,
                                                      float *o0, float *o1, float *o2, int64_t C) {
  const int64_t i = block * TV_SLOT_COLS + threadIdx.x % TV_SLOT_COLS;
  double tot[1];
  const int64_t s = i / C, c = i - s * C;
  float *o = (s == 0) ? o0 : (s == 1 ? o1 : o2);
  tv_slot_sums<1>(partial, nblk, sc, 0, i, i < sc && o != nullptr, tot);   // a plane without an
  if (threadIdx.x >= TV_SLOT_COLS || i >= sc) return;                      // output was not written
  if (o) o[c] = (float)tot[0];
}
__global__ __launch_bounds__(256) void reduce_partials_kernel(int64_t nblk, int64_t sc,
                                                              const float *__restrict__ partial,
                                                              float *o0, float *o1, float *o2,
                                                              int64_t C) {
  reduce_partials_block(blockIdx.x, nblk, sc, partial, o0, o1, o2, C);
}
// What follows the fused backward pass, as ONE launch (both are a few microseconds of work and
// independent of each other): workgroups [0, cols_blocks) write the [del | der | 0] columns,
// the rest add up the bias-gradient partials.
__global__ __launch_bounds__(256) void gat_bwd_tail_kernel(AggArgs a, int64_t cols_blocks,
                                                           int64_t nblk, int64_t sc, float *dbias,
                                                           int64_t C) {
  if ((int64_t)blockIdx.x < cols_blocks)
    gat_bwd_score_cols_block(a, blockIdx.x);
  else
    reduce_partials_block((int64_t)blockIdx.x - cols_blocks, nblk, sc, a.partial, nullptr, nullptr,
                          dbias, C);
}

// launch KERNEL<CH, HT, ...> with CH = 4 or 8 float4 chunks per lane and HT = 1, 2 or MAXH heads
#define TV_GAT_LAUNCH(KERNEL, GRID, BLOCK, LDSB, STREAM, ARGS, ...)                                \
  do {                                                                                           \
    const int ht_ = (ARGS).H == 1 ? 1 : ((ARGS).H == 2 ? 2 : MAXH);                              \
    if ((ARGS).c4 > 256) {                                                                       \
      if (ht_ == 1) hipLaunchKernelGGL((KERNEL<8, 1 __VA_ARGS__>), GRID, BLOCK, LDSB, STREAM, ARGS);      \
      else if (ht_ == 2) hipLaunchKernelGGL((KERNEL<8, 2 __VA_ARGS__>), GRID, BLOCK, LDSB, STREAM, ARGS); \
      else hipLaunchKernelGGL((KERNEL<8, MAXH __VA_ARGS__>), GRID, BLOCK, LDSB, STREAM, ARGS);            \
    } else {                                                                                     \
      if (ht_ == 1) hipLaunchKernelGGL((KERNEL<4, 1 __VA_ARGS__>), GRID, BLOCK, LDSB, STREAM, ARGS);      \
      else if (ht_ == 2) hipLaunchKernelGGL((KERNEL<4, 2 __VA_ARGS__>), GRID, BLOCK, LDSB, STREAM, ARGS); \
      else hipLaunchKernelGGL((KERNEL<4, MAXH __VA_ARGS__>), GRID, BLOCK, LDSB, STREAM, ARGS);            \
    }                                                                                            \
  } while (0)

// the windowed forward form needs more than 64 KB of LDS per workgroup: raised once per device
template <int HT, bool STATS, bool EPILOGUE>
void launch_lds(dim3 grid, dim3 block, size_t ldsb, hipStream_t s, const AggArgs &a) {
  static bool raised[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !raised[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gat_aggregate_fwd_lds_kernel<HT, STATS, EPILOGUE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised[dev] = true;
  }
  hipLaunchKernelGGL((gat_aggregate_fwd_lds_kernel<HT, STATS, EPILOGUE>), grid, block, ldsb, s, a);
}
// fewest 32-row windows for which the windowed form is taken (one workgroup per CU: below a
// round of the chip the per-row kernel spreads the rows over more waves)
int64_t gat_lds_min_windows() {
  const char *e = getenv("TV_GAT_LDS_MIN_WINDOWS");
  return e ? atoll(e) : 512;
}

// workgroups of the statistics-free aggregation (the row loop strides over the grid)
int64_t tv_gat_max_blocks() { return 2048; }

int check_shape(const char *name, int64_t rows_per_rep, int64_t n_rep, int64_t heads,
                int64_t feats) {
  TV_REQUIRE(rows_per_rep >= 0 && n_rep >= 1, "%s: bad row counts", name);
  TV_REQUIRE(heads >= 1 && heads <= MAXH, "%s: heads must be in [1,%d]: got %lld", name, MAXH,
             (long long)heads);
  TV_REQUIRE(feats >= 4 && feats % 4 == 0, "%s: feats must be a positive multiple of 4: got %lld",
             name, (long long)feats);
  TV_REQUIRE(heads * feats <= 2048, "%s: heads*feats must be <= 2048: got %lld", name,
             (long long)(heads * feats));
  return TV_OK;
}


// row strides of ft and of el / er: 0 = dense; ft rows are read 16 bytes per lane
int check_strides(const char *name, int64_t heads, int64_t feats, const float *ft,
                  int64_t *ft_stride, int64_t *score_stride) {
  if (*ft_stride == 0) *ft_stride = heads * feats;
  if (*score_stride == 0) *score_stride = heads;
  TV_REQUIRE(*ft_stride >= heads * feats && *ft_stride % 4 == 0, "%s: ft stride %lld must be a multiple of 4, >= heads*feats",
             name, (long long)*ft_stride);
  TV_REQUIRE((((uintptr_t)ft) & 15) == 0, "%s: ft must be 16-byte aligned", name);
  TV_REQUIRE(*score_stride >= heads, "%s: score stride %lld < heads", name, (long long)*score_stride);
  return TV_OK;
}

// W_ext = [W ; A_l W ; A_r W ; 0]: the fc weight [H*F, K] with 2H more rows (padded with zero
// rows to a multiple of 4), row H*F + h =
// sum_f attn_l[h, f] W[h*F + f, :] (then attn_r), so that el = ft . attn_l = x . (W^T attn_l)
// and er come out of the fc product itself as 2H more columns and the rows of ft are not read
// again to form them.  A workgroup owns 16 columns k of one head, 64 row groups deep (64-byte
// row segments; with 64 columns per workgroup a 1000-wide layer filled 32 of the 256 CUs and the
// launch took 13 us of every layer's forward pass).
constexpr int SW_COLS = 16, SW_GROUPS = 64;
__device__ __forceinline__ void gat_fc_score_weight_block(int bx, int by, int gx, int gy, int H, int F,
                                                          int64_t K, const float *__restrict__ W,
                                                          const float *__restrict__ al,
                                                          const float *__restrict__ ar,
                                                          float *__restrict__ Wx,
                                                          float *__restrict__ w_max) {
  __shared__ float part[2][SW_GROUPS][SW_COLS];
  float wmx = 0.f;                                 // max |entry| of what this thread writes
  const int kx = threadIdx.x % SW_COLS, fg = threadIdx.x / SW_COLS;
  const int h = by;
  const int64_t k = (int64_t)bx * SW_COLS + kx;
  float sl = 0.f, sr = 0.f;
  if (k < K) {
    // eight rows in flight per thread: the kernel is bound by the latency of its loads
    const float *wp = W + (int64_t)h * F * K + k;
    float *xp = Wx + (int64_t)h * F * K + k;
    int f = fg;
    for (; f + 7 * SW_GROUPS < F; f += 8 * SW_GROUPS) {
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = wp[(int64_t)(f + SW_GROUPS * u) * K];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xp[(int64_t)(f + SW_GROUPS * u) * K] = w[u];
        wmx = fmaxf(wmx, fabsf(w[u]));
        sl = fmaf(al[h * F + f + SW_GROUPS * u], w[u], sl);
        sr = fmaf(ar[h * F + f + SW_GROUPS * u], w[u], sr);
      }
    }
    for (; f < F; f += SW_GROUPS) {
      const float w = wp[(int64_t)f * K];
      xp[(int64_t)f * K] = w;
      wmx = fmaxf(wmx, fabsf(w));
      sl = fmaf(al[h * F + f], w, sl);
      sr = fmaf(ar[h * F + f], w, sr);
    }
  }
  part[0][fg][kx] = sl;
  part[1][fg][kx] = sr;
  __syncthreads();
  if (fg < 2 && k < K) {                           // row groups summed in order
    float t = 0.f;
    for (int g = 0; g < SW_GROUPS; ++g) t += part[fg][g][kx];
    Wx[((int64_t)H * F + fg * H + h) * K + k] = t;
    wmx = fmaxf(wmx, fabsf(t));
  }
  if (w_max) {                                     // for the fp16 GEMM forms (tv_gemm_f32_ex, state 2)
    __shared__ float wave_max[16];
    wmx = tv_wave_max(wmx);
    const int nblocks = gx * gy, block = bx + gx * by;
    if (nblocks <= 256) {
      // a slot per workgroup, the rest zeroed here: plain stores, nothing to clear beforehand
      if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = wmx;
      __syncthreads();
      if (threadIdx.x == 0) {
        float m = 0.f;
        for (int w = 0; w < 16; ++w) m = fmaxf(m, wave_max[w]);
        w_max[block] = m;
      }
      if (block == 0)
        for (int sl = nblocks + (int)threadIdx.x; sl < 256; sl += (int)blockDim.x) w_max[sl] = 0.f;
    } else if ((threadIdx.x & 63) == 0) {          // (the host cleared the slots)
      atomicMax(reinterpret_cast<unsigned *>(w_max) + (block & 255), __float_as_uint(wmx));
    }
  }
  // rows that pad 2H up to a multiple of 4 (rows of [ft | el | er] stay 16-byte aligned)
  const int pad = ((2 * H + 3) & ~3) - 2 * H;
  if (h == 0 && fg >= 2 && fg < 2 + pad && k < K) Wx[((int64_t)H * F + 2 * H + (fg - 2)) * K + k] = 0.f;
}
__global__ __launch_bounds__(1024) void gat_fc_score_weight_kernel(int H, int F, int64_t K,
                                                                   const float *__restrict__ W,
                                                                   const float *__restrict__ al,
                                                                   const float *__restrict__ ar,
                                                                   float *__restrict__ Wx,
                                                                   float *__restrict__ w_max) {
  gat_fc_score_weight_block((int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, H, F, K, W,
                            al, ar, Wx, w_max);
}
// the same for the layers of a stack in one launch: blockIdx.z picks the layer
constexpr int SW_BATCH = 16;
struct ScoreWeightItem {
  const float *W, *al, *ar;
  float *Wx, *w_max;
  int64_t K;
  int H, F;
};
struct ScoreWeightBatch { ScoreWeightItem item[SW_BATCH]; };
__global__ __launch_bounds__(1024) void gat_fc_score_weight_batch_kernel(ScoreWeightBatch batch) {
  const ScoreWeightItem &it = batch.item[blockIdx.z];
  const int gx = (int)((it.K + SW_COLS - 1) / SW_COLS), gy = it.H;
  if ((int)blockIdx.x >= gx || (int)blockIdx.y >= gy) return;
  gat_fc_score_weight_block((int)blockIdx.x, (int)blockIdx.y, gx, gy, it.H, it.F, it.K, it.W, it.al, it.ar,
                            it.Wx, it.w_max);
}

// Backward of tv_gat_fc_score_weight.  dWx [H*F + pad4(2H), K] is the weight gradient of the fc
// product with the extended weight ([dft | del | der]^T x); rows H*F + h and H*F + H + h are the
// gradients of the derived rows A_l W and A_r W:
//   dW[h*F+f, :] = dWx[h*F+f, :] + attn_l[h,f] dWx[H*F+h, :] + attn_r[h,f] dWx[H*F+H+h, :]  (in place)
//   dattn_l[h,f] = <W[h*F+f, :], dWx[H*F+h, :]>,  dattn_r likewise.
// One wave per row of W.
__global__ __launch_bounds__(256) void gat_fc_score_weight_bwd_kernel(int H, int F, int64_t K,
                                                                      const float *__restrict__ W,
                                                                      const float *__restrict__ al,
                                                                      const float *__restrict__ ar,
                                                                      float *__restrict__ dWx,
                                                                      float *__restrict__ dal,
                                                                      float *__restrict__ dar) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= (int64_t)H * F) return;
  const int h = (int)(row / F);
  const float a_l = al[row], a_r = ar[row];
  const float *gl = dWx + ((int64_t)H * F + h) * K, *gr = dWx + ((int64_t)H * F + H + h) * K;
  float sl = 0.f, sr = 0.f;
  for (int64_t k = lane; k < K; k += 64) {
    const float w = W[row * K + k], l = gl[k], r = gr[k];
    sl = fmaf(w, l, sl);
    sr = fmaf(w, r, sr);
    dWx[row * K + k] += a_l * l + a_r * r;
  }
  sl = tv_wave_sum(sl);
  sr = tv_wave_sum(sr);
  if (lane == 0) { dal[row] = sl; dar[row] = sr; }
}

}  // namespace

extern "C" int tv_gat_scores(int64_t rows, int64_t heads, int64_t feats, const float *ft,
                             const float *attn_l, const float *attn_r, float *el, float *er,
                             void *stream) {
  if (int rc = check_shape("tv_gat_scores", rows, 1, heads, feats)) return rc;
  if (rows == 0) return TV_OK;
  TV_REQUIRE(ft && attn_l && attn_r && el && er, "tv_gat_scores: null pointer");
  int64_t blocks = tv_cdiv(rows, 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gat_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     rows, (int)heads, (int)feats, ft, attn_l, attn_r, el, er);
  TV_LAUNCH_CHECK("tv_gat_scores");
  return TV_OK;
}

extern "C" int tv_gat_fc_score_weight(int64_t heads, int64_t feats, int64_t in_feats,
                                      const float *weight, const float *attn_l,
                                      const float *attn_r, float *weight_ext, float *weight_ext_max,
                                      void *stream) {
  if (int rc = check_shape("tv_gat_fc_score_weight", 0, 1, heads, feats)) return rc;
  TV_REQUIRE(in_feats >= 1, "tv_gat_fc_score_weight: in_feats must be positive");
  TV_REQUIRE(weight && attn_l && attn_r && weight_ext, "tv_gat_fc_score_weight: null pointer");
  const dim3 grid((unsigned)tv_cdiv(in_feats, SW_COLS), (unsigned)heads), block(1024);
  if (weight_ext_max && (int64_t)grid.x * grid.y > 256)
    TV_HIP(hipMemsetAsync(weight_ext_max, 0, 256 * sizeof(float), (hipStream_t)stream));
  hipLaunchKernelGGL(gat_fc_score_weight_kernel, grid, block, 0, (hipStream_t)stream, (int)heads,
                     (int)feats, in_feats, weight, attn_l, attn_r, weight_ext, weight_ext_max);
  TV_LAUNCH_CHECK("tv_gat_fc_score_weight");
  return TV_OK;
}

extern "C" int tv_gat_fc_score_weight_batch(int64_t count, const int64_t *heads, const int64_t *feats,
                                            const int64_t *in_feats, const float *const *weight,
                                            const float *const *attn_l, const float *const *attn_r,
                                            float *const *weight_ext, float *const *weight_ext_max,
                                            void *stream) {
  TV_REQUIRE(count >= 0 && count <= SW_BATCH, "tv_gat_fc_score_weight_batch: at most %d layers per call",
             SW_BATCH);
  if (count == 0) return TV_OK;
  TV_REQUIRE(heads && feats && in_feats && weight && attn_l && attn_r && weight_ext && weight_ext_max,
             "tv_gat_fc_score_weight_batch: null array");
  ScoreWeightBatch batch = {};
  int64_t gx = 0, gy = 0;
  for (int64_t i = 0; i < count; ++i) {
    if (int rc = check_shape("tv_gat_fc_score_weight_batch", 0, 1, heads[i], feats[i])) return rc;
    TV_REQUIRE(in_feats[i] >= 1, "tv_gat_fc_score_weight_batch: in_feats must be positive");
    TV_REQUIRE(weight[i] && attn_l[i] && attn_r[i] && weight_ext[i] && weight_ext_max[i],
               "tv_gat_fc_score_weight_batch: null pointer");
    const int64_t bx = tv_cdiv(in_feats[i], SW_COLS);
    // (the maxima go to one slot per workgroup: no atomics, nothing to clear)
    TV_REQUIRE(bx * heads[i] <= 256, "tv_gat_fc_score_weight_batch: layer %lld needs %lld workgroups (> 256)",
               (long long)i, (long long)(bx * heads[i]));
    ScoreWeightItem &it = batch.item[i];
    it.W = weight[i]; it.al = attn_l[i]; it.ar = attn_r[i]; it.Wx = weight_ext[i]; it.w_max = weight_ext_max[i];
    it.K = in_feats[i]; it.H = (int)heads[i]; it.F = (int)feats[i];
    if (bx > gx) gx = bx;
    if (heads[i] > gy) gy = heads[i];
  }
  hipLaunchKernelGGL(gat_fc_score_weight_batch_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)count),
                     dim3(1024), 0, (hipStream_t)stream, batch);
  TV_LAUNCH_CHECK("tv_gat_fc_score_weight_batch");
  return TV_OK;
}

extern "C" int tv_gat_fc_score_weight_bwd(int64_t heads, int64_t feats, int64_t in_feats,
                                          const float *weight, const float *attn_l,
                                          const float *attn_r, float *dweight_ext, float *dattn_l,
                                          float *dattn_r, void *stream) {
  if (int rc = check_shape("tv_gat_fc_score_weight_bwd", 0, 1, heads, feats)) return rc;
  TV_REQUIRE(in_feats >= 1, "tv_gat_fc_score_weight_bwd: in_feats must be positive");
  TV_REQUIRE(weight && attn_l && attn_r && dweight_ext && dattn_l && dattn_r,
             "tv_gat_fc_score_weight_bwd: null pointer");
  hipLaunchKernelGGL(gat_fc_score_weight_bwd_kernel, dim3((unsigned)tv_cdiv(heads * feats, 4)),
                     dim3(256), 0, (hipStream_t)stream, (int)heads, (int)feats, in_feats, weight,
                     attn_l, attn_r, dweight_ext, dattn_l, dattn_r);
  TV_LAUNCH_CHECK("tv_gat_fc_score_weight_bwd");
  return TV_OK;
}

extern "C" int tv_gat_aggregate_fwd(int64_t rows_per_rep, int64_t n_rep, int64_t heads,
                                    int64_t feats, const int64_t *indptr, const int64_t *indices,
                                    const float *ft, const float *el, const float *er,
                                    const float *bias, float *out, float *alpha,
                                    float *stats_partial, const float *stats_row_weight,
                                    const float *ep_scale, const float *ep_shift, float ep_slope,
                                    int64_t ft_stride, int64_t score_stride, float *out_max,
                                    void *stream) {
  if (int rc = check_shape("tv_gat_aggregate_fwd", rows_per_rep, n_rep, heads, feats)) return rc;
  TV_REQUIRE(!out_max || ep_scale, "tv_gat_aggregate_fwd: out_max goes with the eval epilogue");
  if (int rc = check_strides("tv_gat_aggregate_fwd", heads, feats, ft, &ft_stride, &score_stride)) return rc;
  const int64_t rows = rows_per_rep * n_rep;
  if (rows == 0) return TV_OK;
  TV_REQUIRE(indptr && indices && ft && el && er && bias && out,
             "tv_gat_aggregate_fwd: null pointer");
  TV_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr),
             "tv_gat_aggregate_fwd: ep_scale and ep_shift go together");
  TV_REQUIRE(!(stats_partial && ep_scale),
             "tv_gat_aggregate_fwd: statistics and the fused eval epilogue are exclusive");
  AggArgs a = {};
  a.rows_per_rep = rows_per_rep; a.n_rep = n_rep; a.rows = rows;
  a.H = (int)heads; a.F = (int)feats; a.c4 = (int)(heads * feats / 4);
  a.indptr = indptr; a.indices = indices; a.ft = ft; a.el = el; a.er = er; a.bias = bias;
  a.out = out; a.alpha = alpha; a.stats_partial = stats_partial;
  a.row_weight = stats_row_weight;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.ep_slope = ep_slope;
  a.ld_ft = ft_stride; a.ld_e = score_stride;
  a.out_max = out_max;
  if (out_max) TV_HIP(hipMemsetAsync(out_max, 0, 256 * sizeof(float), (hipStream_t)stream));
  // num_edges of one replica is only needed to index alpha; read back is avoided by
  // passing it through indptr on the host side of the caller: alpha is indexed with
  // rep * num_edges, and num_edges = indptr[rows_per_rep] is a device value.  With one
  // replica the term vanishes; with several the caller must pass alpha == NULL.
  TV_REQUIRE(!(alpha && n_rep > 1),
             "tv_gat_aggregate_fwd: alpha is only kept for a single replica (training)");
  a.num_edges = 0;
  hipStream_t s = (hipStream_t)stream;
  const bool big = a.c4 > 256;
  a.stats_slots = tv_stat_blocks_host(rows);
  // the windowed form: [ft | el | er] rows of at most 1024 floats, enough windows for the chip
  const bool columns = el == ft + heads * feats && er == el + heads && score_stride == ft_stride;
  const int64_t windows = tv_cdiv(rows, LDS_ROWS);
  // (the statistics variant of a training step keeps 32 more registers per lane and spills in
  // this form: 117 against 52 us per launch - it stays on the per-row kernel)
  if (columns && !stats_partial && ft_stride <= 1024 && windows >= gat_lds_min_windows()) {
    int64_t blocks = windows < 256 * LDS_WGS_PER_CU ? windows : 256 * LDS_WGS_PER_CU;
    const dim3 grid((unsigned)blocks), block(64 * LDS_WAVES);
    const size_t ldsb = sizeof(float) * (size_t)LDS_ROWS * ft_stride;
    const int ht = heads == 1 ? 1 : (heads == 2 ? 2 : MAXH);
#define TV_GAT_LDS_LAUNCH(ST, EP)                                                                   \
    do {                                                                                           \
      if (ht == 1) launch_lds<1, ST, EP>(grid, block, ldsb, s, a);                                 \
      else if (ht == 2) launch_lds<2, ST, EP>(grid, block, ldsb, s, a);                            \
      else launch_lds<MAXH, ST, EP>(grid, block, ldsb, s, a);                                      \
    } while (0)
    if (ep_scale) TV_GAT_LDS_LAUNCH(false, true);
    else TV_GAT_LDS_LAUNCH(false, false);
#undef TV_GAT_LDS_LAUNCH
    TV_LAUNCH_CHECK("tv_gat_aggregate_fwd(windowed)");
    return TV_OK;
  }
  if (stats_partial) {
    const dim3 grid((unsigned)tv_stat_blocks_host(rows)), block(64 * TV_STAT_WAVES);
    const size_t ldsb = sizeof(float) * 2 * heads * feats;
    TV_GAT_LAUNCH(gat_aggregate_fwd_kernel, grid, block, ldsb, s, a, , true, false);
  } else {
    int64_t blocks = tv_cdiv(rows, TV_STAT_WAVES);
    if (blocks > tv_gat_max_blocks()) blocks = tv_gat_max_blocks();
    const dim3 grid((unsigned)blocks), block(64 * TV_STAT_WAVES);
    if (ep_scale) {
      TV_GAT_LAUNCH(gat_aggregate_fwd_kernel, grid, block, 0, s, a, , false, true);
    } else {
      TV_GAT_LAUNCH(gat_aggregate_fwd_kernel, grid, block, 0, s, a, , false, false);
    }
  }
  TV_LAUNCH_CHECK("tv_gat_aggregate_fwd");
  return TV_OK;
}

static inline int64_t pad4(int64_t floats) { return (floats + 3) & ~3ll; }

extern "C" int64_t tv_gat_aggregate_bwd_workspace_bytes(int64_t rows_per_rep, int64_t n_rep,
                                                        int64_t heads, int64_t feats,
                                                        int64_t num_edges) {
  const int64_t rows = rows_per_rep * n_rep;
  const int64_t nblk = tv_stat_blocks_host(rows);
  // dpre | der | partial, each segment a multiple of 4 floats: the partials are written 16 bytes
  // at a time
  return (int64_t)sizeof(float) *
         (pad4(n_rep * num_edges * heads) + pad4(rows * heads) + nblk * 3 * heads * feats);
}

extern "C" int tv_gat_aggregate_bwd(int64_t rows_per_rep, int64_t n_rep, int64_t heads,
                                    int64_t feats, int64_t num_edges, const int64_t *indptr,
                                    const int64_t *indices, const int64_t *rev, const float *ft,
                                    const float *el, const float *er, const float *attn_l,
                                    const float *attn_r, const float *alpha, const float *dout,
                                    float *dft, float *dattn_l, float *dattn_r, float *dbias,
                                    int64_t ft_stride, int64_t score_stride, int64_t dft_stride,
                                    int score_grad_columns, float *dft_max, void *workspace,
                                    int64_t workspace_bytes, void *stream) {
  if (int rc = check_shape("tv_gat_aggregate_bwd", rows_per_rep, n_rep, heads, feats)) return rc;
  TV_REQUIRE(!dft_max || score_grad_columns,
             "tv_gat_aggregate_bwd: dft_max goes with the score-gradient columns");
  if (int rc = check_strides("tv_gat_aggregate_bwd", heads, feats, ft, &ft_stride, &score_stride)) return rc;
  if (dft_stride == 0) dft_stride = heads * feats;
  TV_REQUIRE(!score_grad_columns || dft_stride >= heads * feats + ((2 * heads + 3) & ~3ll),
             "tv_gat_aggregate_bwd: dft rows too short for the score-gradient columns");
  TV_REQUIRE(dft_stride >= heads * feats && dft_stride % 4 == 0 && (((uintptr_t)dft) & 15) == 0,
             "tv_gat_aggregate_bwd: dft rows must be 16-byte aligned, stride >= heads*feats");
  TV_REQUIRE(n_rep == 1, "tv_gat_aggregate_bwd: the backward pass runs on a single replica");
  const int64_t rows = rows_per_rep * n_rep;
  if (rows == 0) return TV_OK;
  TV_REQUIRE(indptr && indices && rev && ft && el && er && alpha && dout && dft && dbias,
             "tv_gat_aggregate_bwd: null pointer");
  TV_REQUIRE(score_grad_columns || (attn_l && attn_r && dattn_l && dattn_r),
             "tv_gat_aggregate_bwd: null attention vector / gradient");
  const int64_t need =
      tv_gat_aggregate_bwd_workspace_bytes(rows_per_rep, n_rep, heads, feats, num_edges);
  if (!workspace || workspace_bytes < need) {
    tv_set_error("tv_gat_aggregate_bwd: workspace of %lld bytes needed, %lld given",
                 (long long)need, (long long)workspace_bytes);
    return TV_ERR_WORKSPACE;
  }
  const int64_t nblk = tv_stat_blocks_host(rows);
  const int64_t HF = heads * feats;
  AggArgs a = {};
  a.rows_per_rep = rows_per_rep; a.n_rep = n_rep; a.rows = rows; a.num_edges = num_edges;
  a.H = (int)heads; a.F = (int)feats; a.c4 = (int)(HF / 4);
  a.indptr = indptr; a.indices = indices; a.rev = rev; a.ft = ft; a.el = el; a.er = er;
  a.attn_l = attn_l; a.attn_r = attn_r; a.alpha_in = alpha; a.dout = dout; a.dft = dft;
  a.ld_ft = ft_stride; a.ld_e = score_stride; a.ld_dft = dft_stride;
  a.dpre = (float *)workspace;
  a.der = a.dpre + pad4(n_rep * num_edges * heads);
  a.partial = a.der + pad4(rows * heads);
  TV_REQUIRE(((uintptr_t)workspace & 15) == 0, "tv_gat_aggregate_bwd: workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  a.out_max = dft_max;
  if (dft_max) TV_HIP(hipMemsetAsync(dft_max, 0, 256 * sizeof(float), s));
  const bool big = a.c4 > 256;
  const dim3 grid((unsigned)nblk), block(64 * TV_STAT_WAVES);
  if (score_grad_columns) {
    TV_GAT_LAUNCH(gat_bwd_fused_kernel, grid, block, sizeof(float) * HF, s, a);
    TV_LAUNCH_CHECK("tv_gat_aggregate_bwd(fused)");
    const int64_t cols_blocks = tv_cdiv(4 * rows, 256);
    hipLaunchKernelGGL(gat_bwd_tail_kernel, dim3((unsigned)(cols_blocks + tv_slot_grid(3 * HF))),
                       dim3(256), 0, s, a, cols_blocks, nblk, 3 * HF, dbias, HF);
    TV_LAUNCH_CHECK("tv_gat_aggregate_bwd(score columns + bias gradient)");
    return TV_OK;
  } else {
    // same grid for both passes: they write their column partials into slot blockIdx.x
    TV_GAT_LAUNCH(gat_bwd_edge_kernel, grid, block, sizeof(float) * HF, s, a);
    TV_LAUNCH_CHECK("tv_gat_aggregate_bwd(edge)");
    TV_GAT_LAUNCH(gat_bwd_node_kernel, grid, block, sizeof(float) * 2 * HF, s, a);
    TV_LAUNCH_CHECK("tv_gat_aggregate_bwd(node)");
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(tv_slot_grid(3 * HF)), dim3(256), 0, s,
                     nblk, 3 * HF, a.partial, score_grad_columns ? nullptr : dattn_l,
                     score_grad_columns ? nullptr : dattn_r, dbias, HF);
  TV_LAUNCH_CHECK("tv_gat_aggregate_bwd(reduce)");
  return TV_OK;
}
