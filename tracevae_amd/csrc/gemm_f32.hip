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
  int a_vec, b_vec;          // 16-byte loads allowed for the operand
};

__device__ __forceinline__ float4 load4_guard(const float *base, int64_t row, int64_t nrows,
                                              int64_t col, int64_t ncols, int64_t ld, int vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < nrows && col < ncols) {
    const float *p = base + row * ld + col;
    if (vec_ok && col + 3 < ncols) {
      v = *reinterpret_cast<const float4 *>(p);
    } else {
      v.x = p[0];
      if (col + 1 < ncols) v.y = p[1];
      if (col + 2 < ncols) v.z = p[2];
      if (col + 3 < ncols) v.w = p[3];
    }
  }
  return v;
}

// Operand stored [X][K] (K contiguous): thread t loads float4 k-quads of 4 rows.
//   tile element (x, k): x = (t>>3) + 32*it, k = (t&7)*4 + j
// Operand stored [K][X] (X contiguous): thread t loads float4 x-quads of 4 k-rows.
//   tile element (x, k): x = (t&31)*4 + j, k = (t>>5) + 8*it
template <bool KMAJOR>
__device__ __forceinline__ void load_tile(float4 (&r)[4], const float *base, int64_t x0,
                                          int64_t xlim, int64_t k0, int64_t klim, int64_t ld,
                                          int vec_ok, int t) {
  if (!KMAJOR) {
    const int64_t k = k0 + (t & 7) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      r[it] = load4_guard(base, x0 + (t >> 3) + 32 * it, xlim, k, klim, ld, vec_ok);
  } else {
    const int64_t x = x0 + (t & 31) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      r[it] = load4_guard(base, k0 + (t >> 5) + 8 * it, klim, x, xlim, ld, vec_ok);
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void store_tile(const float4 (&r)[4], float *s, int t) {
  if (!KMAJOR) {
    const int k = (t & 7) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int x = (t >> 3) + 32 * it;
      s[(k + 0) * LDS_STRIDE + x] = r[it].x;
      s[(k + 1) * LDS_STRIDE + x] = r[it].y;
      s[(k + 2) * LDS_STRIDE + x] = r[it].z;
      s[(k + 3) * LDS_STRIDE + x] = r[it].w;
    }
  } else {
    const int x = (t & 31) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int k = (t >> 5) + 8 * it;
      *reinterpret_cast<float4 *>(&s[k * LDS_STRIDE + x]) = r[it];
    }
  }
}

template <bool A_KMAJOR, bool B_KMAJOR>
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

  float4 ra[4], rb[4];
  float *As = smem, *Bs = smem + 2 * TILE_FLOATS;

  if (nt > 0) {
    load_tile<A_KMAJOR>(ra, g.A, m0, g.M, kbeg, kend, g.lda, g.a_vec, t);
    load_tile<B_KMAJOR>(rb, g.B, n0, g.N, kbeg, kend, g.ldb, g.b_vec, t);
    store_tile<A_KMAJOR>(ra, As, t);
    store_tile<B_KMAJOR>(rb, Bs, t);
  }
  __syncthreads();

  const int kh = lane >> 5;   // which of the two k of an MFMA step this lane feeds
  const int xl = lane & 31;
  for (int64_t it = 0; it < nt; ++it) {
    const int cur = (int)(it & 1);
    const bool more = (it + 1 < nt);
    if (more) {
      const int64_t k0 = kbeg + (it + 1) * BK;
      load_tile<A_KMAJOR>(ra, g.A, m0, g.M, k0, kend, g.lda, g.a_vec, t);
      load_tile<B_KMAJOR>(rb, g.B, n0, g.N, k0, kend, g.ldb, g.b_vec, t);
    }
    const float *as = As + cur * TILE_FLOATS + wm * 64 + xl;
    const float *bs = Bs + cur * TILE_FLOATS + wn * 64 + xl;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = as[(kk + kh) * LDS_STRIDE];
      const float a1 = as[(kk + kh) * LDS_STRIDE + 32];
      const float b0 = bs[(kk + kh) * LDS_STRIDE];
      const float b1 = bs[(kk + kh) * LDS_STRIDE + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      store_tile<A_KMAJOR>(ra, As + (cur ^ 1) * TILE_FLOATS, t);
      store_tile<B_KMAJOR>(rb, Bs + (cur ^ 1) * TILE_FLOATS, t);
    }
    __syncthreads();
  }

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool split = g.splits > 1;
  float *out = split ? (g.slabs + sp * g.M * g.N) : g.C;
  const int64_t ldo = split ? g.N : g.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + xl;
      if (col >= g.N) continue;
      const float bv = (!split && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= g.M) continue;
        float v = acc[i][j][r];
        if (!split) {
          if (g.row_scale) v *= g.row_scale[row];
          v += bv;
          if (g.accumulate) v += out[row * ldo + col];
        }
        out[row * ldo + col] = v;
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
  g.a_vec = (lda % 4 == 0) && (((uintptr_t)a & 15) == 0);
  g.b_vec = (ldb % 4 == 0) && (((uintptr_t)b & 15) == 0);
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
  const bool ak = trans_a != 0, bk = trans_b == 0;
  if (!ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, g);
  else if (!ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, g);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, g);
  TV_LAUNCH_CHECK("tv_gemm_f32");
  if (g.splits > 1) {
    int64_t blocks = tv_cdiv(m * n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g);
    TV_LAUNCH_CHECK("tv_gemm_f32(split-k reduce)");
  }
  return TV_OK;
}
