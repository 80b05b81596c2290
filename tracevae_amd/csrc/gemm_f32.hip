// Dense layers on the fp32 matrix cores of gfx950.
//
//   C[M,N] = row_scale[m] * (op(A)[M,K] . op(B)[K,N]) + bias[n]      (fp32 in, fp32 accumulate)
//
// v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 fma chain, which is what the 1e-4
// parity bound on losses / scores needs (no reduced-precision path is used).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave owns a
// 64x64 block = 2x2 MFMA tiles of 32x32 -> 64 accumulator registers), BK = 32.
// Both operands are staged through LDS in [k][m] order (row stride 132 floats), so the MFMA
// operand read is one conflict-free ds_read_b32 per lane (lanes 0-31 read 32 consecutive
// floats of row k, lanes 32-63 of row k+1).  Global -> register -> LDS with two LDS buffers:
// the loads of tile t+1 are in flight while tile t is multiplied, one barrier per tile.
// Workgroup ids are remapped so that the workgroups sharing an XCD (ids equal mod 8) walk a
// contiguous range of tiles, n fastest: an A panel is then fetched into one L2 only.
// Bound: MFMA (157.3 TFLOP/s fp32 dense peak); algorithmic flops = 2*M*N*K.
//
// A reduction over many rows into a small output (weight gradients: M,N ~ 1000, K = rows)
// is split over K into slabs and summed by a second kernel in slab order (deterministic).
#include "tv_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_STRIDE = 132;              // floats; 528 B rows keep 16-B alignment
constexpr int TILE_FLOATS = BK * LDS_STRIDE; // one operand, one buffer

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float *A, *B;
  float *C;
  const float *bias, *row_scale;
  int64_t M, N, K, lda, ldb, ldc;
  int64_t tiles_m, tiles_n;
  int64_t splits, k_chunk;   // k_chunk is a multiple of BK
  float *slabs;              // [splits, M, N] when splits > 1
  int accumulate;
  int a_vec, b_vec;          // 16-byte loads allowed for the operand (host-side dispatch)
};

// Tile staging.  One operand tile is 128 (x) by 32 (k) floats, 16 floats per thread.
// Every load is unconditional from a clamped (always valid) address and zeroed by a
// select afterwards, so the main loop has no divergent branches.
//   VEC  (16-B loads; needs ld % 4 == 0, 16-B aligned base, contiguous extent % 4 == 0)
//     [X][K] storage: float4 = 4 consecutive k of row x = (t>>3) + 32*it, k = (t&7)*4
//     [K][X] storage: float4 = 4 consecutive x of row k = (t>>5) + 8*it,  x = (t&31)*4
//   scalar (any ld / alignment): element e = t + 256*i
//     [X][K] storage: k = e & 31,  x = e >> 5     (32 consecutive floats of a row)
//     [K][X] storage: x = e & 127, k = e >> 7     (128 consecutive floats of a k-row)
template <bool KMAJOR, bool VEC>
struct Stager {
  float r[16];
  unsigned okmask;   // validity of the staged loads; applied when the tile is written to LDS,
                     // so that nothing consumes a load result before the MFMA phase
  const float *base;
  int64_t ld, x0, xlim;
  int t;

  __device__ __forceinline__ void init(const float *b, int64_t ld_, int64_t x0_, int64_t xlim_,
                                       int t_) {
    base = b; ld = ld_; x0 = x0_; xlim = xlim_; t = t_;
  }

  __device__ __forceinline__ void load(int64_t k0, int64_t klim) {
    okmask = 0;
    if (VEC) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        int64_t x, k;
        if (!KMAJOR) { x = x0 + (t >> 3) + 32 * it; k = k0 + (t & 7) * 4; }
        else         { x = x0 + (t & 31) * 4;       k = k0 + (t >> 5) + 8 * it; }
        const bool ok = (x < xlim) && (k < klim);
        const int64_t xc = ok ? x : 0, kc = ok ? k : 0;
        const float *p = KMAJOR ? (base + kc * ld + xc) : (base + xc * ld + kc);
        const float4 v = *reinterpret_cast<const float4 *>(p);
        r[4 * it + 0] = v.x; r[4 * it + 1] = v.y; r[4 * it + 2] = v.z; r[4 * it + 3] = v.w;
        okmask |= ok ? (1u << it) : 0u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int64_t x, k;
        if (!KMAJOR) { k = k0 + (t & 31);  x = x0 + (t >> 5) + 8 * i; }
        else         { x = x0 + (t & 127); k = k0 + (t >> 7) + 2 * i; }
        const bool ok = (x < xlim) && (k < klim);
        const int64_t xc = ok ? x : 0, kc = ok ? k : 0;
        r[i] = KMAJOR ? base[kc * ld + xc] : base[xc * ld + kc];
        okmask |= ok ? (1u << i) : 0u;
      }
    }
  }

  __device__ __forceinline__ void store(float *s) const {
    if (VEC) {
      if (!KMAJOR) {
        const int k = (t & 7) * 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int x = (t >> 3) + 32 * it;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            s[(k + j) * LDS_STRIDE + x] = ((okmask >> it) & 1u) ? r[4 * it + j] : 0.f;
        }
      } else {
        const int x = (t & 31) * 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int k = (t >> 5) + 8 * it;
          const bool ok = (okmask >> it) & 1u;
          *reinterpret_cast<float4 *>(&s[k * LDS_STRIDE + x]) =
              make_float4(ok ? r[4 * it] : 0.f, ok ? r[4 * it + 1] : 0.f,
                          ok ? r[4 * it + 2] : 0.f, ok ? r[4 * it + 3] : 0.f);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = ((okmask >> i) & 1u) ? r[i] : 0.f;
        if (!KMAJOR) s[(t & 31) * LDS_STRIDE + (t >> 5) + 8 * i] = v;
        else         s[((t >> 7) + 2 * i) * LDS_STRIDE + (t & 127)] = v;
      }
    }
  }
};

template <bool A_KMAJOR, bool B_KMAJOR, bool A_VEC, bool B_VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __align__(16) float smem[4 * TILE_FLOATS];  // A0 A1 B0 B1
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware, bijective remap of the workgroup id (ids equal mod 8 share an XCD)
  const int64_t nwg = gridDim.x;
  const int64_t bid = blockIdx.x;
  const int64_t q = nwg >> 3, rr = nwg & 7;
  const int64_t xcd = bid & 7, idx = bid >> 3;
  const int64_t L = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  const int64_t tn = L % g.tiles_n;
  const int64_t tm = (L / g.tiles_n) % g.tiles_m;
  const int64_t sp = L / (g.tiles_n * g.tiles_m);

  const int64_t m0 = tm * BM, n0 = tn * BN;
  const int64_t kbeg = sp * g.k_chunk;
  const int64_t kend = (kbeg + g.k_chunk < g.K) ? (kbeg + g.k_chunk) : g.K;
  const int64_t nt = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<A_KMAJOR, A_VEC> sa;
  Stager<B_KMAJOR, B_VEC> sb;
  sa.init(g.A, g.lda, m0, g.M, t);
  sb.init(g.B, g.ldb, n0, g.N, t);
  float *As = smem, *Bs = smem + 2 * TILE_FLOATS;

  if (nt > 0) {
    sa.load(kbeg, kend);
    sb.load(kbeg, kend);
    sa.store(As);
    sb.store(Bs);
  }
  __syncthreads();

  const int kh = lane >> 5;   // which of the two k of an MFMA step this lane feeds
  const int xl = lane & 31;
  for (int64_t it = 0; it < nt; ++it) {
    const int cur = (int)(it & 1);
    const bool more = (it + 1 < nt);
    if (more) {
      const int64_t k0 = kbeg + (it + 1) * BK;
      sa.load(k0, kend);
      sb.load(k0, kend);
    }
    const float *as = As + cur * TILE_FLOATS + wm * 64 + xl + kh * LDS_STRIDE;
    const float *bs = Bs + cur * TILE_FLOATS + wn * 64 + xl + kh * LDS_STRIDE;
    // operand fragments are read one k-step ahead of the MFMAs that use them
    float fa[2][2], fb[2][2];
    fa[0][0] = as[0]; fa[0][1] = as[32]; fb[0][0] = bs[0]; fb[0][1] = bs[32];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int c = (kk >> 1) & 1, nx = c ^ 1;
      if (kk + 2 < BK) {
        fa[nx][0] = as[(kk + 2) * LDS_STRIDE];
        fa[nx][1] = as[(kk + 2) * LDS_STRIDE + 32];
        fb[nx][0] = bs[(kk + 2) * LDS_STRIDE];
        fb[nx][1] = bs[(kk + 2) * LDS_STRIDE + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
      // pin the emitted order: the LDS reads of the next k-step go out before this step's
      // four MFMAs, so their latency hides under 256 cycles of matrix work
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    if (more) {
      sa.store(As + (cur ^ 1) * TILE_FLOATS);
      sb.store(Bs + (cur ^ 1) * TILE_FLOATS);
    }
    __syncthreads();
  }

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool split = g.splits > 1;
  float *out = split ? (g.slabs + sp * g.M * g.N) : g.C;
  const int64_t ldo = split ? g.N : g.ldc;
  const bool plain = split || (!g.row_scale && !g.accumulate);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + xl;
      const bool colok = col < g.N;
      const float bv = (!split && g.bias && colok) ? g.bias[col] : 0.f;
      const int64_t rbase = m0 + wm * 64 + i * 32 + 4 * kh;
      if (plain) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
          if (colok && row < g.M) out[row * ldo + col] = acc[i][j][r] + bv;
        }
      } else {
        for (int r = 0; r < 16; ++r) {
          const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
          if (colok && row < g.M) {
            float v = acc[i][j][r];
            if (g.row_scale) v *= g.row_scale[row];
            v += bv;
            if (g.accumulate) v += out[row * ldo + col];
            out[row * ldo + col] = v;
          }
        }
      }
    }
  }
}

// second stage of split-K: out = epilogue(sum_s slab[s]) in slab order
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g) {
  const int64_t total = g.M * g.N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / g.N, col = e % g.N;
    float v = 0.f;
    for (int64_t s = 0; s < g.splits; ++s) v += g.slabs[s * total + e];
    if (g.row_scale) v *= g.row_scale[row];
    if (g.bias) v += g.bias[col];
    float *o = g.C + row * g.ldc + col;
    if (g.accumulate) v += *o;
    *o = v;
  }
}

int64_t choose_splits(int64_t m, int64_t n, int64_t k) {
  const int64_t tiles = tv_cdiv(m, BM) * tv_cdiv(n, BN);
  if (tiles >= 256 || k < 8 * BK) return 1;
  int64_t s = tv_cdiv(512, tiles);
  const int64_t kmax = k / (4 * BK);  // at least 4 k-tiles per split
  if (s > kmax) s = kmax;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}

}  // namespace

extern "C" int64_t tv_gemm_f32_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const int64_t s = choose_splits(m, n, k);
  return s > 1 ? s * m * n * (int64_t)sizeof(float) : 0;
}

extern "C" int tv_gemm_f32(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k,
                           const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                           int64_t ldc, const float *bias, const float *row_scale,
                           int accumulate, void *workspace, int64_t workspace_bytes,
                           void *stream) {
  TV_REQUIRE(m >= 0 && n >= 0 && k >= 0, "tv_gemm_f32: negative size");
  if (m == 0 || n == 0) return TV_OK;
  TV_REQUIRE(a && b && c, "tv_gemm_f32: null operand");
  TV_REQUIRE(lda >= (trans_a ? m : k), "tv_gemm_f32: lda %lld too small", (long long)lda);
  TV_REQUIRE(ldb >= (trans_b ? k : n), "tv_gemm_f32: ldb %lld too small", (long long)ldb);
  TV_REQUIRE(ldc >= n, "tv_gemm_f32: ldc %lld too small", (long long)ldc);
  TV_REQUIRE(k > 0, "tv_gemm_f32: k must be positive");

  GemmArgs g;
  g.A = a; g.B = b; g.C = c; g.bias = bias; g.row_scale = row_scale;
  g.M = m; g.N = n; g.K = k; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tiles_m = tv_cdiv(m, BM); g.tiles_n = tv_cdiv(n, BN);
  g.splits = choose_splits(m, n, k);
  g.k_chunk = tv_cdiv(tv_cdiv(k, g.splits), BK) * BK;
  g.splits = tv_cdiv(k, g.k_chunk);
  g.slabs = nullptr;
  g.accumulate = accumulate;
  // 16-B loads need aligned rows and a contiguous extent that is a multiple of 4
  g.a_vec = (lda % 4 == 0) && (((uintptr_t)a & 15) == 0) && ((trans_a ? m : k) % 4 == 0);
  g.b_vec = (ldb % 4 == 0) && (((uintptr_t)b & 15) == 0) && ((trans_b ? k : n) % 4 == 0);
  if (g.splits > 1) {
    const int64_t need = g.splits * m * n * (int64_t)sizeof(float);
    if (!workspace || workspace_bytes < need) {
      tv_set_error("tv_gemm_f32: workspace of %lld bytes needed, %lld given", (long long)need,
                   (long long)workspace_bytes);
      return TV_ERR_WORKSPACE;
    }
    g.slabs = (float *)workspace;
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t nwg = g.tiles_m * g.tiles_n * g.splits;
  TV_REQUIRE(nwg < (1ll << 31), "tv_gemm_f32: grid too large");
  const dim3 grid((unsigned)nwg), block(256);
  const int variant = (trans_a != 0 ? 8 : 0) | (trans_b == 0 ? 4 : 0) | (g.a_vec ? 2 : 0) |
                      (g.b_vec ? 1 : 0);
#define TV_GEMM_CASE(V, AK, BK_, AV, BV)                                                        \
  case V: hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, AV, BV>), grid, block, 0, s, g); break;
  switch (variant) {
    TV_GEMM_CASE(0, false, false, false, false) TV_GEMM_CASE(1, false, false, false, true)
    TV_GEMM_CASE(2, false, false, true, false)  TV_GEMM_CASE(3, false, false, true, true)
    TV_GEMM_CASE(4, false, true, false, false)  TV_GEMM_CASE(5, false, true, false, true)
    TV_GEMM_CASE(6, false, true, true, false)   TV_GEMM_CASE(7, false, true, true, true)
    TV_GEMM_CASE(8, true, false, false, false)  TV_GEMM_CASE(9, true, false, false, true)
    TV_GEMM_CASE(10, true, false, true, false)  TV_GEMM_CASE(11, true, false, true, true)
    TV_GEMM_CASE(12, true, true, false, false)  TV_GEMM_CASE(13, true, true, false, true)
    TV_GEMM_CASE(14, true, true, true, false)   TV_GEMM_CASE(15, true, true, true, true)
  }
#undef TV_GEMM_CASE
  TV_LAUNCH_CHECK("tv_gemm_f32");
  if (g.splits > 1) {
    int64_t blocks = tv_cdiv(m * n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g);
    TV_LAUNCH_CHECK("tv_gemm_f32(split-k reduce)");
  }
  return TV_OK;
}
