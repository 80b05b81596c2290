// Dense layers on the fp32 matrix cores of gfx950.
//
//   C[M,N] = row_scale[m] * (op(A)[M,K] . op(B)[K,N]) + bias[n]      (fp32 in, fp32 accumulate)
//
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain, which is what the 1e-4 parity bound on
// losses / scores needs (no reduced-precision path is used).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave owns a
// 64x64 block = 2x2 MFMA tiles of 32x32 -> 64 accumulator registers), BK = 32, two workgroups
// per CU.  An operand tile is staged in LDS in the orientation it has in memory, so staging is
// a 16-byte load and a 16-byte LDS store with no transposition:
//   row-major ([x][k] storage): LDS [x][k], row stride 36 floats; a lane's MFMA operands for one
//     8-deep k-block are ONE ds_read_b128 (4 consecutive k of its row, conflict-free at stride 36)
//   k-major ([k][x] storage):   LDS [k][x], row stride 132 floats; one ds_read_b32 per k-step
// In both, lanes 0-31 hold k = 8*kb + s and lanes 32-63 hold k = 8*kb + 4 + s for step s of
// k-block kb: an MFMA step multiplies that k pair (a fixed summation order, the same for A and B).
// Pipeline (one barrier per tile): barrier; write the registers holding tile t+1 into the other
// LDS buffer; issue the global loads of tile t+2; multiply tile t.  Loads therefore have a whole
// tile of matrix work to land, and the LDS stores sit behind the first MFMAs of the tile.
// Edges: out-of-range rows/columns read a clamped (valid) address and only ever reach
// accumulators that are not stored; the one partial k-tile is loaded through a masked path.
// Workgroup ids are remapped so that the workgroups sharing an XCD (ids equal mod 8) walk a
// contiguous range of tiles, n fastest: an A panel is then fetched into one L2 only.
// Bound: MFMA (157.3 TFLOP/s fp32 dense peak); algorithmic flops = 2*M*N*K.
//
// A reduction over many rows into a small output (weight gradients: M,N ~ 1000, K = rows)
// is split over K into slabs and summed by a second kernel in slab order (deterministic).
#include "tv_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int SX = 36;                        // row stride of an [x][k] tile (floats)
constexpr int SK = 132;                       // row stride of a [k][x] tile
constexpr int TILE_FLOATS = (BM * SX > BK * SK) ? BM * SX : BK * SK;  // one operand, one buffer

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float *A, *B;
  float *C;
  const float *bias, *row_scale;
  int64_t M, N, K, lda, ldb, ldc;
  int64_t tiles_m, tiles_n;
  int64_t splits, k_chunk;   // k_chunk is a multiple of BK
  float *slabs;              // [splits, M, N] when splits > 1
  // tail split: output tiles >= tail_first (the last, partial round of workgroups over the
  // chip) are each computed by tail_splits workgroups over K/tail_splits, into packed
  // 128x128 slabs [tile - tail_first][split], summed by gemm_tail_reduce_kernel
  int64_t tail_first, tail_splits, tail_kchunk;
  float *tail_slabs;
  int accumulate;
  int a_vec, b_vec;          // 16-byte loads allowed for the operand (host-side dispatch)
  int c_vec;                 // 16-byte stores allowed for the output (ldc, N, base all % 4)
  // operand prologue (tv_gemm_f32_act): v -> lrelu(v * act_scale[f] + act_shift[f]) on the way
  // into LDS, f = the operand's feature index (k for a row-major A, n for a k-major B)
  const float *act_scale, *act_shift;
  float act_slope;
};

// Staging of one operand tile: 128 (x) by 32 (k) floats, 16 floats per thread.
//   KX  = the operand is stored [k][x] (x contiguous), else [x][k] (k contiguous)
//   VEC = 16-B loads (needs ld % 4 == 0, 16-B aligned base, contiguous extent % 4 == 0),
//         else 4-B loads (any ld / alignment)
// thread -> element maps (x, k local to the tile):
//   [x][k] VEC:    float4 it: x = (t>>3) + 32*it, k = (t&7)*4 ..+3
//   [k][x] VEC:    float4 it: k = (t>>5) + 8*it,  x = (t&31)*4 ..+3
//   [x][k] scalar: element i: x = (t>>5) + 8*i,   k = t&31
//   [k][x] scalar: element i: k = (t>>7) + 2*i,   x = t&127
//   ACT = the operand goes through lrelu(v * scale[f] + shift[f]) before it is stored to LDS;
//         f = k for an [x][k] operand (scale/shift follow the k-tiles), f = x for a [k][x] one
//         (fixed per thread).  Elements of the zero-filled part of a partial k-tile stay zero
//         for f = k; for f = x they become lrelu(shift), harmless because the other operand's
//         zero-filled k rows multiply them.
template <bool KX, bool VEC, bool ACT>
struct Stage {
  static constexpr int NL = VEC ? 4 : 16;
  static constexpr int NF = VEC ? 4 : 1;      // features a thread touches
  float r[16];
  float sc[NF], sh[NF];
  const float *act_scale, *act_shift;
  float slope;
  unsigned fpos;                              // feature index of this thread's first element
  // element offset of load i from the tile origin = min(off0 + i*step, offmax): consecutive
  // loads of a thread are `step` apart (a multiple of ld), and offmax clamps an out-of-range x
  // (rows of an [x][k] operand) onto the last valid row -- such an x only feeds accumulators
  // that are never stored, it just has to read a valid element
  unsigned off0, offmax, step;
  const float *tile;  // origin of the current tile (wave-uniform)
  int64_t ld;
  int t;

  __device__ __forceinline__ int klocal(int i) const {
    if (VEC) return KX ? (t >> 5) + 8 * i : (t & 7) * 4;
    return KX ? (t >> 7) + 2 * i : (t & 31);
  }
  __device__ __forceinline__ int xlocal(int i) const {
    if (VEC) return KX ? (t & 31) * 4 : (t >> 3) + 32 * i;
    return KX ? (t & 127) : (t >> 5) + 8 * i;
  }
  __device__ __forceinline__ unsigned offset(int i) const {
    const unsigned o = off0 + (unsigned)i * step;
    return KX ? o : (o < offmax ? o : offmax);
  }

  __device__ __forceinline__ void init(const float *base, int64_t ld_, int64_t x0, int64_t xlim,
                                       int64_t k0, int t_) {
    ld = ld_; t = t_;
    tile = KX ? base + k0 * ld + x0 : base + x0 * ld + k0;
    if (KX) {
      int x = xlocal(0);
      if (x0 + x + (VEC ? 3 : 0) >= xlim) x = VEC ? 0 : (int)(xlim - 1 - x0);
      off0 = (unsigned)(klocal(0) * ld + x);
      step = (unsigned)((VEC ? 8 : 2) * ld);
      offmax = 0xffffffffu;
    } else {
      off0 = (unsigned)(xlocal(0) * ld + klocal(0));
      step = (unsigned)((VEC ? 32 : 8) * ld);
      offmax = (unsigned)((xlim - 1 - x0) * ld + klocal(0));
    }
  }

  __device__ __forceinline__ void init_act(const float *scale, const float *shift, float slope_,
                                           int64_t x0, int64_t xlim, int64_t k0) {
    if (!ACT) return;
    act_scale = scale; act_shift = shift; slope = slope_;
    if (KX) {
      // the features are this thread's columns: the same (clamped) x as in init()
      int x = xlocal(0);
      if (x0 + x + (VEC ? 3 : 0) >= xlim) x = VEC ? 0 : (int)(xlim - 1 - x0);
      fpos = (unsigned)(x0 + x);
      load_act(NF);
    } else {
      fpos = (unsigned)(k0 + klocal(0));
    }
  }

  // scale / shift of features fpos .. fpos+NF-1; the first `valid` of them exist
  __device__ __forceinline__ void load_act(int valid) {
    if (VEC && valid >= NF) {
      const float4 a = *reinterpret_cast<const float4 *>(act_scale + fpos);
      const float4 b = *reinterpret_cast<const float4 *>(act_shift + fpos);
      sc[0] = a.x; sc[NF > 1 ? 1 : 0] = a.y; sc[NF > 2 ? 2 : 0] = a.z; sc[NF > 3 ? 3 : 0] = a.w;
      sh[0] = b.x; sh[NF > 1 ? 1 : 0] = b.y; sh[NF > 2 ? 2 : 0] = b.z; sh[NF > 3 ? 3 : 0] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const bool ok = j < valid;
        sc[j] = ok ? act_scale[fpos + j] : 0.f;
        sh[j] = ok ? act_shift[fpos + j] : 0.f;
      }
    }
  }

  __device__ __forceinline__ void advance() {
    tile += KX ? BK * ld : BK;
    if (ACT && !KX) fpos += BK;
  }

  // a full k-tile: unconditional loads
  __device__ __forceinline__ void load() {
    if (ACT && !KX) load_act(NF);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (VEC) {
        const float4 v = *reinterpret_cast<const float4 *>(tile + offset(i));
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
      } else {
        r[i] = tile[offset(i)];
      }
    }
  }

  // the partial last k-tile: k >= rem reads the tile's first k instead and is zeroed
  __device__ __forceinline__ void load_tail(int rem) {
    if (ACT && !KX) load_act(rem - klocal(0));
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int kl = klocal(i);
      const bool ok = kl < rem;
      const unsigned o = ok ? offset(i) : offset(i) - (KX ? (unsigned)(kl * ld) : (unsigned)kl);
      if (VEC) {
        const float4 v = *reinterpret_cast<const float4 *>(tile + o);
        r[4 * i] = ok ? v.x : 0.f; r[4 * i + 1] = ok ? v.y : 0.f;
        r[4 * i + 2] = ok ? v.z : 0.f; r[4 * i + 3] = ok ? v.w : 0.f;
      } else {
        const float v = tile[o];
        r[i] = ok ? v : 0.f;
      }
    }
  }

  // lrelu(y) = max(y, slope*y) for 0 <= slope <= 1 (checked on the host): three VALU ops per
  // element, written on float2 so that the fma and the multiply become packed-math instructions
  __device__ __forceinline__ void act2(float &a, float &b, int ja, int jb) const {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v = {a, b};
    const f32x2 s2 = {sc[ja], sc[jb]}, h2 = {sh[ja], sh[jb]};
    const f32x2 y = v * s2 + h2;
    const f32x2 ys = y * slope;
    a = fmaxf(y.x, ys.x);
    b = fmaxf(y.y, ys.y);
  }

  // the prologue, in registers.  Called after the multiply that follows the loads, so that the
  // arithmetic sits in the shadow of the MFMAs instead of between the barrier and the LDS stores
  __device__ __forceinline__ void transform() {
    transform_half(0);
    transform_half(1);
  }
  __device__ __forceinline__ void transform_half(int h) {
    if (!ACT) return;
#pragma unroll
    for (int i = 8 * h; i < 8 * h + 8; i += 2)
      act2(r[i], r[i + 1], VEC ? (i & 3) : 0, VEC ? ((i + 1) & 3) : 0);
  }

  __device__ __forceinline__ void store(float *s) const {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int pos = KX ? klocal(i) * SK + xlocal(i) : xlocal(i) * SX + klocal(i);
      if (VEC) *reinterpret_cast<float4 *>(&s[pos]) = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
      else s[pos] = r[i];
    }
  }
};

// MFMA operands of k-block kb for the two 32-wide sub-tiles of a wave: f[sub][step]
template <bool KX>
__device__ __forceinline__ void read_fragments(const float *s, int w64, int lane, int kb,
                                               float (&f)[2][4]) {
  const int xl = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    if (!KX) {
      const float4 v = *reinterpret_cast<const float4 *>(&s[(w64 + sub * 32 + xl) * SX + 8 * kb + 4 * kh]);
      f[sub][0] = v.x; f[sub][1] = v.y; f[sub][2] = v.z; f[sub][3] = v.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) f[sub][q] = s[(8 * kb + 4 * kh + q) * SK + w64 + sub * 32 + xl];
    }
  }
}

// acc += A-tile . B-tile over `nkb` k-blocks of 8 (FULL: all four, fragments of block kb+1 are
// read while block kb is multiplied)
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// `hook(kb)` runs after the MFMAs of k-block kb have been issued (the steady-state loop puts the
// operand prologue of the tile in flight there, in the shadow of the MFMA pipeline)
template <bool A_KX, bool B_KX, bool FULL, typename Hook = NoHook>
__device__ __forceinline__ void multiply_tile(const float *as, const float *bs, int wm64, int wn64,
                                              int lane, int nkb, f32x16 (&acc)[2][2],
                                              Hook hook = Hook()) {
  float fa[2][2][4], fb[2][2][4];
  read_fragments<A_KX>(as, wm64, lane, 0, fa[0]);
  read_fragments<B_KX>(bs, wn64, lane, 0, fb[0]);
#pragma unroll
  for (int kb = 0; kb < BK / 8; ++kb) {
    if (!FULL && kb >= nkb) break;
    const int c = kb & 1;
    if (kb + 1 < BK / 8 && (FULL || kb + 1 < nkb)) {
      read_fragments<A_KX>(as, wm64, lane, kb + 1, fa[c ^ 1]);
      read_fragments<B_KX>(bs, wn64, lane, kb + 1, fb[c ^ 1]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0][s], fb[c][0][s], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0][s], fb[c][1][s], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1][s], fb[c][0][s], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1][s], fb[c][1][s], acc[1][1], 0, 0, 0);
    }
    hook(kb);
  }
}

// ACT: 0 = plain, 1 = prologue on A (row-major, !A_KX), 2 = prologue on B (k-major, B_KX)
template <bool A_KX, bool B_KX, bool A_VEC, bool B_VEC, int ACT = 0>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
  static_assert(ACT == 0 || (ACT == 1 && !A_KX) || (ACT == 2 && B_KX), "prologue orientation");
  __shared__ __align__(16) float smem[4 * TILE_FLOATS];  // A0 A1 B0 B1
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // Workgroup id -> (output tile, k-split).  Ids below `remapped` are permuted so that the
  // workgroups sharing an XCD (ids equal mod 8) walk a contiguous range of tiles, n fastest
  // (bijective for any count); the tail-split units keep dispatch order so that the last,
  // small units spread over all XCDs.  Everything here is wave-uniform: keep it in SGPRs.
  const int tail_units = g.tail_splits > 1;
  const int nwg = tail_units ? (int)g.tail_first : (int)gridDim.x;
  const int bid = blockIdx.x;
  int tile, sp;
  int64_t kchunk;
  const bool tail = tail_units && bid >= nwg;
  if (tail) {
    const int u = bid - nwg;
    tile = nwg + u / (int)g.tail_splits;
    sp = u % (int)g.tail_splits;
    kchunk = g.tail_kchunk;
  } else {
    const int q = nwg >> 3, rr = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int L = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    const int tiles = (int)(g.tiles_n * g.tiles_m);
    tile = L % tiles;
    sp = L / tiles;
    kchunk = g.k_chunk;
  }
  tile = __builtin_amdgcn_readfirstlane(tile);
  sp = __builtin_amdgcn_readfirstlane(sp);
  const int tn = __builtin_amdgcn_readfirstlane(tile % (int)g.tiles_n);
  const int tm = __builtin_amdgcn_readfirstlane(tile / (int)g.tiles_n);

  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t kbeg = sp * kchunk;
  const int64_t kend = (kbeg + kchunk < g.K) ? (kbeg + kchunk) : g.K;
  const int64_t klen = kend > kbeg ? kend - kbeg : 0;
  const int ntf = __builtin_amdgcn_readfirstlane((int)(klen / BK));          // full k-tiles
  const int rem = __builtin_amdgcn_readfirstlane((int)(klen - (int64_t)ntf * BK));
  const int nt = ntf + (rem ? 1 : 0);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stage<A_KX, A_VEC, ACT == 1> sa;
  Stage<B_KX, B_VEC, ACT == 2> sb;
  sa.init(g.A, g.lda, m0, g.M, kbeg, t);
  sb.init(g.B, g.ldb, n0, g.N, kbeg, t);
  sa.init_act(g.act_scale, g.act_shift, g.act_slope, m0, g.M, kbeg);
  sb.init_act(g.act_scale, g.act_shift, g.act_slope, n0, g.N, kbeg);
  float *As = smem, *Bs = smem + 2 * TILE_FLOATS;

  // the operand prologue runs after the multiply that follows the loads
#define TV_POST_LOAD() do { sa.transform(); sb.transform(); } while (0)
  if (nt > 0) {
    if (ntf > 0) { sa.load(); sb.load(); } else { sa.load_tail(rem); sb.load_tail(rem); }
    TV_POST_LOAD();
    sa.store(As);
    sb.store(Bs);
  }
  if (nt > 1) {
    sa.advance(); sb.advance();
    if (ntf > 1) { sa.load(); sb.load(); } else { sa.load_tail(rem); sb.load_tail(rem); }
    TV_POST_LOAD();
  }

  // steady state: tiles t, t+1 and t+2 are all full k-tiles, nothing is conditional
  int it = 0;
  for (; it + 2 < ntf; ++it) {
    const int cur = it & 1;
    __syncthreads();
    sa.store(As + (cur ^ 1) * TILE_FLOATS);
    sb.store(Bs + (cur ^ 1) * TILE_FLOATS);
    sa.advance(); sb.advance();
    sa.load(); sb.load();
    // keep the loads of tile t+2 ahead of the multiply of tile t: left alone, the scheduler
    // sinks them below the MFMAs (shorter live ranges) and the next iteration then waits for
    // them right after the barrier
    __builtin_amdgcn_sched_barrier(0);
    // the prologue of tile t+2 (its loads were issued before this multiply) goes after the
    // third and fourth k-block, fenced so that it stays between the MFMA groups
    multiply_tile<A_KX, B_KX, true>(As + cur * TILE_FLOATS, Bs + cur * TILE_FLOATS, wm * 64, wn * 64,
                                    lane, BK / 8, acc, [&](int kb) {
                                      if (ACT != 0 && kb >= 2) {
                                        __builtin_amdgcn_sched_barrier(0);
                                        sa.transform_half(kb - 2);
                                        sb.transform_half(kb - 2);
                                        __builtin_amdgcn_sched_barrier(0);
                                      }
                                    });
  }
  // the last (up to three) tiles: stop staging, the partial k-tile goes through the masked path
  for (; it < nt; ++it) {
    const int cur = it & 1;
    __syncthreads();
    if (it + 1 < nt) {
        sa.store(As + (cur ^ 1) * TILE_FLOATS);
      sb.store(Bs + (cur ^ 1) * TILE_FLOATS);
    }
    if (it + 2 < nt) {
      sa.advance(); sb.advance();
      if (it + 2 < ntf) { sa.load(); sb.load(); } else { sa.load_tail(rem); sb.load_tail(rem); }
    }
    // only the k-blocks (8 deep) that hold data: all four, fewer in the partial tile
    multiply_tile<A_KX, B_KX, false>(As + cur * TILE_FLOATS, Bs + cur * TILE_FLOATS, wm * 64, wn * 64,
                                     lane, it < ntf ? BK / 8 : (rem + 7) >> 3, acc);
    if (it + 2 < nt) TV_POST_LOAD();
  }
#undef TV_POST_LOAD

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int kh = lane >> 5, xl = lane & 31;
  if (tail) {
    float *slab = g.tail_slabs + ((int64_t)(tile - nwg) * g.tail_splits + sp) * (BM * BN);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(wm * 64 + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2)) * BN + wn * 64 + j * 32 + xl] =
              acc[i][j][r];
    return;
  }
  const bool split = g.splits > 1;
  float *out = split ? (g.slabs + (int64_t)sp * g.M * g.N) : g.C;
  const int64_t ldo = split ? g.N : g.ldc;
  const bool plain = split || (!g.row_scale && !g.accumulate);
  if (plain && g.c_vec && m0 + BM <= g.M) {
    // all 128 rows of the tile exist and rows can be written 16 bytes per lane: the wave
    // transposes its 64x64 block through LDS (the tile buffers are free now) and stores whole
    // 256-byte row segments - 16 store instructions per lane instead of 64 scattered dwords.
    // Row stride 72 floats: the two 32-lane halves of an accumulator register (rows r, r+4)
    // land 32 banks apart.
    constexpr int ES = 72;
    static_assert(4 * 64 * ES <= 4 * TILE_FLOATS, "epilogue staging fits the tile buffers");
    __syncthreads();                                // every wave is done with the last k-tile
    float *ws = smem + wave * (64 * ES);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ws[(i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2)) * ES + j * 32 + xl] = acc[i][j][r];
    const int c = (lane & 15) * 4;
    const int64_t col = n0 + wn * 64 + c;
    if (col < g.N) {                                // N % 4 == 0: the four columns exist together
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!split && g.bias) bv = make_float4(g.bias[col], g.bias[col + 1], g.bias[col + 2], g.bias[col + 3]);
      float *o = out + (m0 + wm * 64 + (lane >> 4)) * ldo + col;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(&ws[(q * 4 + (lane >> 4)) * ES + c]);
        *reinterpret_cast<float4 *>(o + (int64_t)q * 4 * ldo) =
            make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
      }
    }
    return;
  }
  if (plain && m0 + BM <= g.M) {
    // all 128 rows of the tile exist: stores are only predicated on the column of the lane
    float *o = out + (m0 + wm * 64 + 4 * kh) * ldo + n0 + wn * 64 + xl;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + xl;
      if (col < g.N) {
        const float bv = (!split && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            o[(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo + j * 32] = acc[i][j][r] + bv;
      }
    }
    return;
  }
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

// second stage of the tail split: C tile = epilogue(sum_s slab[tile][s]) in split order.
// One workgroup per 32 rows of a tail tile; a thread owns 4 consecutive columns.
__global__ __launch_bounds__(256) void gemm_tail_reduce_kernel(GemmArgs g) {
  const int64_t rel = blockIdx.x >> 2;
  const int64_t tile = g.tail_first + rel;
  const int64_t tn = tile % g.tiles_n, tm = tile / g.tiles_n;
  const int t = threadIdx.x;
  const int c4 = (t & 31) * 4;
  const float *slab = g.tail_slabs + rel * g.tail_splits * (BM * BN);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = (blockIdx.x & 3) * 32 + (t >> 5) + 8 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t s = 0; s < g.tail_splits; ++s) {
      const float4 p = *reinterpret_cast<const float4 *>(&slab[s * (BM * BN) + rl * BN + c4]);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const int64_t row = tm * BM + rl;
    if (row >= g.M) continue;
    const float rs = g.row_scale ? g.row_scale[row] : 1.0f;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = tn * BN + c4 + j;
      if (col < g.N) {
        float w = e[j] * rs;
        if (g.bias) w += g.bias[col];
        float *o = g.C + row * g.ldc + col;
        if (g.accumulate) w += *o;
        *o = w;
      }
    }
  }
}

// workgroups resident at once: two per CU (LDS-bound)
int64_t resident_workgroups() {
  static int64_t slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    slots = 2 * (int64_t)cus;
  }
  return slots;
}

struct Plan {
  int64_t splits, k_chunk;                          // whole-problem split-K
  int64_t tail_first, tail_splits, tail_kchunk;     // tail split (mutually exclusive with it)
  int64_t workspace_bytes;
};

int64_t choose_splits(int64_t m, int64_t n, int64_t k);

Plan make_plan(int64_t m, int64_t n, int64_t k) {
  Plan p;
  const int64_t tiles = tv_cdiv(m, BM) * tv_cdiv(n, BN);
  p.splits = choose_splits(m, n, k);
  p.k_chunk = tv_cdiv(tv_cdiv(k, p.splits), BK) * BK;
  p.splits = tv_cdiv(k, p.k_chunk);
  p.tail_first = tiles; p.tail_splits = 1; p.tail_kchunk = p.k_chunk;
  p.workspace_bytes = p.splits > 1 ? p.splits * m * n * (int64_t)sizeof(float) : 0;
  const int64_t slots = resident_workgroups();
  static const bool tail_enabled = [] { const char *e = getenv("TV_GEMM_TAIL_SPLIT"); return !(e && e[0] == '0'); }();
  if (tail_enabled && p.splits == 1 && tiles > slots && tiles < (1ll << 30)) {
    const int64_t tail = tiles % slots;
    if (tail > 0) {
      int64_t ts = slots / tail;
      const int64_t kmax = k / (4 * BK);             // at least 4 k-tiles per unit
      if (ts > kmax) ts = kmax;
      if (ts > 8) ts = 8;
      if (ts >= 2) {
        const int64_t kc = tv_cdiv(tv_cdiv(k, ts), BK) * BK;
        ts = tv_cdiv(k, kc);
        if (ts >= 2) {
          p.tail_first = tiles - tail; p.tail_splits = ts; p.tail_kchunk = kc;
          p.workspace_bytes = tail * ts * (int64_t)(BM * BN * sizeof(float));
        }
      }
    }
  }
  return p;
}

int64_t choose_splits(int64_t m, int64_t n, int64_t k) {
  const int64_t tiles = tv_cdiv(m, BM) * tv_cdiv(n, BN);
  if (tiles >= 256 || k < 4 * BK) return 1;
  int64_t s = tv_cdiv(512, tiles);
  // a workgroup alone on its CU pays the full memory latency per k-tile, so few-tile problems
  // are split deep: at least TV_GEMM_MIN_SPLIT_TILES (default 2) k-tiles per split
  static const int64_t min_tiles = [] {
    const char *e = getenv("TV_GEMM_MIN_SPLIT_TILES");
    return (e && atoll(e) > 0) ? atoll(e) : 2ll;
  }();
  const int64_t kmax = k / (min_tiles * BK);
  if (s > kmax) s = kmax;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}

}  // namespace

extern "C" int64_t tv_gemm_f32_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  return make_plan(m, n, k).workspace_bytes;
}

namespace {
int run_gemm(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float *a,
             int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc, const float *bias,
             const float *row_scale, int accumulate, int act_operand, const float *act_scale,
             const float *act_shift, float act_slope, void *workspace, int64_t workspace_bytes,
             void *stream) {
  TV_REQUIRE(m >= 0 && n >= 0 && k >= 0, "tv_gemm_f32: negative size");
  if (m == 0 || n == 0) return TV_OK;
  TV_REQUIRE(a && b && c, "tv_gemm_f32: null operand");
  TV_REQUIRE(lda >= (trans_a ? m : k), "tv_gemm_f32: lda %lld too small", (long long)lda);
  TV_REQUIRE(ldb >= (trans_b ? k : n), "tv_gemm_f32: ldb %lld too small", (long long)ldb);
  TV_REQUIRE(ldc >= n, "tv_gemm_f32: ldc %lld too small", (long long)ldc);
  TV_REQUIRE(k > 0, "tv_gemm_f32: k must be positive");
  TV_REQUIRE(lda < (1ll << 24) && ldb < (1ll << 24), "tv_gemm_f32: leading dimension too large");

  GemmArgs g;
  g.A = a; g.B = b; g.C = c; g.bias = bias; g.row_scale = row_scale;
  g.M = m; g.N = n; g.K = k; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tiles_m = tv_cdiv(m, BM); g.tiles_n = tv_cdiv(n, BN);
  const Plan plan = make_plan(m, n, k);
  g.splits = plan.splits; g.k_chunk = plan.k_chunk;
  g.tail_first = plan.tail_first; g.tail_splits = plan.tail_splits; g.tail_kchunk = plan.tail_kchunk;
  g.slabs = nullptr; g.tail_slabs = nullptr;
  g.accumulate = accumulate;
  g.act_scale = act_scale; g.act_shift = act_shift; g.act_slope = act_slope;
  TV_REQUIRE(act_operand >= 0 && act_operand <= 2, "tv_gemm_f32_act: act_operand must be 0, 1 or 2");
  if (act_operand != 0) {
    TV_REQUIRE(act_scale && act_shift, "tv_gemm_f32_act: null scale / shift");
    TV_REQUIRE(act_operand != 1 || !trans_a, "tv_gemm_f32_act: the A prologue needs trans_a = 0");
    TV_REQUIRE(act_operand != 2 || !trans_b, "tv_gemm_f32_act: the B prologue needs trans_b = 0");
    TV_REQUIRE((((uintptr_t)act_scale | (uintptr_t)act_shift) & 15) == 0,
               "tv_gemm_f32_act: scale / shift must be 16-byte aligned");
    TV_REQUIRE(act_slope >= 0.f && act_slope <= 1.f, "tv_gemm_f32_act: slope must be in [0, 1]");
  }
  // 16-B loads need aligned rows and a contiguous extent that is a multiple of 4
  g.a_vec = (lda % 4 == 0) && (((uintptr_t)a & 15) == 0) && ((trans_a ? m : k) % 4 == 0);
  g.b_vec = (ldb % 4 == 0) && (((uintptr_t)b & 15) == 0) && ((trans_b ? k : n) % 4 == 0);
  g.c_vec = (ldc % 4 == 0) && (((uintptr_t)c & 15) == 0) && (n % 4 == 0);
  if (plan.workspace_bytes > 0) {
    const int64_t need = plan.workspace_bytes;
    if (!workspace || workspace_bytes < need) {
      tv_set_error("tv_gemm_f32: workspace of %lld bytes needed, %lld given", (long long)need,
                   (long long)workspace_bytes);
      return TV_ERR_WORKSPACE;
    }
    if (g.splits > 1) {
      g.slabs = (float *)workspace;      // the kernel's output is then a slab: [M, N] rows
      g.c_vec = (n % 4 == 0) && (((uintptr_t)workspace & 15) == 0);
    } else {
      g.tail_slabs = (float *)workspace;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t tiles = g.tiles_m * g.tiles_n;
  const int64_t nwg = g.tail_splits > 1 ? g.tail_first + (tiles - g.tail_first) * g.tail_splits
                                        : tiles * g.splits;
  TV_REQUIRE(nwg < (1ll << 31), "tv_gemm_f32: grid too large");
  const dim3 grid((unsigned)nwg), block(256);
  const int variant = (trans_a != 0 ? 8 : 0) | (trans_b == 0 ? 4 : 0) | (g.a_vec ? 2 : 0) |
                      (g.b_vec ? 1 : 0);
#define TV_GEMM_ACT_CASE(V, AK, BK_, AV, BV, ACT)                                               \
  case V: hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, AV, BV, ACT>), grid, block, 0, s, g); break;
  if (act_operand == 1) {          // A row-major
    switch (variant) {
      TV_GEMM_ACT_CASE(0, false, false, false, false, 1) TV_GEMM_ACT_CASE(1, false, false, false, true, 1)
      TV_GEMM_ACT_CASE(2, false, false, true, false, 1)  TV_GEMM_ACT_CASE(3, false, false, true, true, 1)
      TV_GEMM_ACT_CASE(4, false, true, false, false, 1)  TV_GEMM_ACT_CASE(5, false, true, false, true, 1)
      TV_GEMM_ACT_CASE(6, false, true, true, false, 1)   TV_GEMM_ACT_CASE(7, false, true, true, true, 1)
    }
  } else if (act_operand == 2) {   // B k-major
    switch (variant) {
      TV_GEMM_ACT_CASE(4, false, true, false, false, 2)  TV_GEMM_ACT_CASE(5, false, true, false, true, 2)
      TV_GEMM_ACT_CASE(6, false, true, true, false, 2)   TV_GEMM_ACT_CASE(7, false, true, true, true, 2)
      TV_GEMM_ACT_CASE(12, true, true, false, false, 2)  TV_GEMM_ACT_CASE(13, true, true, false, true, 2)
      TV_GEMM_ACT_CASE(14, true, true, true, false, 2)   TV_GEMM_ACT_CASE(15, true, true, true, true, 2)
    }
  } else
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
#undef TV_GEMM_ACT_CASE
  TV_LAUNCH_CHECK("tv_gemm_f32");
  if (g.splits > 1) {
    int64_t blocks = tv_cdiv(m * n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g);
    TV_LAUNCH_CHECK("tv_gemm_f32(split-k reduce)");
  }
  if (g.tail_splits > 1) {
    hipLaunchKernelGGL(gemm_tail_reduce_kernel, dim3((unsigned)((tiles - g.tail_first) * 4)), dim3(256),
                       0, s, g);
    TV_LAUNCH_CHECK("tv_gemm_f32(tail reduce)");
  }
  return TV_OK;
}
}  // namespace

extern "C" int tv_gemm_f32(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k,
                           const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                           int64_t ldc, const float *bias, const float *row_scale,
                           int accumulate, void *workspace, int64_t workspace_bytes,
                           void *stream) {
  return run_gemm(trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, bias, row_scale, accumulate,
                  0, nullptr, nullptr, 0.f, workspace, workspace_bytes, stream);
}

extern "C" int tv_gemm_f32_act(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k,
                               const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                               int64_t ldc, const float *bias, const float *row_scale,
                               int accumulate, int act_operand, const float *act_scale,
                               const float *act_shift, float act_slope, void *workspace,
                               int64_t workspace_bytes, void *stream) {
  return run_gemm(trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, bias, row_scale, accumulate,
                  act_operand, act_scale, act_shift, act_slope, workspace, workspace_bytes, stream);
}
