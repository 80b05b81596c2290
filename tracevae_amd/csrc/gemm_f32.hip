// Dense layers on the matrix cores of gfx950: fp32 operands, fp32 result, fp32-accurate.
//
//   C[M,N] = row_scale[m] * (op(A)[M,K] . op(B)[K,N]) + bias[n]
//
// Two arithmetic forms, chosen by shape (run_gemm):
//   * the WIDE forms (128 x 256 tiles: the layer products and weight gradients of many rows, which
//     carry the time) split every element into two fp16 planes under a power-of-two operand
//     scale and issue three matrix instructions per product - described where they are defined,
//     "The two wide forms on TWO fp16 planes";
//   * every other shape runs on the producer / consumer kernel below, on three bf16 planes and
//     six matrix instructions per product, which needs no scale.  The rest of this header is about
//     that form.
//
// Every fp32 operand element is split into three bf16 planes x = x0 + x1 + x2 (round-to-nearest
// at each level: 8 + 8 + 8 mantissa bits), and the six partial products of weight >= 2^-16,
//     a0 b0 | a0 b1 + a1 b0 | a0 b2 + a1 b1 + a2 b0,
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate of
// v_mfma_f32_32x32x2_f32, so 16/6 = 2.7x per product).  bf16 products are exact in fp32; a0 b0
// goes to one accumulator and the five correction terms to a second one, added once at the end,
// so the large accumulator is rounded once per 16-deep k-step (63 times at K = 1000 against
// 1000 times in an fp32 fma chain).  Measured on MI355X against float64 the result is 2-4x
// CLOSER than the exact-fp32 MFMA chain on normal, wide-dynamic-range, cancelling and
// all-positive data (tools/bf16x3_numerics.hip; tests/test_gpu_ops.py holds the bound against
// an fp32 GEMM; the dropped terms a1 b2, a2 b1, a2 b2 are below 2^-24 relative).  An infinite
// operand gives NaN (inf - inf in the split) where an fp32 chain gives inf; NaN stays NaN.
//
// Tiling: 128x128 output tile, BK = 32, one 512-thread workgroup per CU:
//   waves 0-3  CONSUMERS, 2x2, each owns a 64x64 block = 2x2 MFMA tiles of 32x32, twice (the two
//              accumulators) -> 128 accumulator registers; per k-tile 24 ds_read_b128 feed 48 MFMAs
//   waves 4-7  PRODUCERS: global loads two tiles ahead -> operand prologue -> split -> the bf16
//              planes of the next tile into the other LDS buffer
// One of each kind sits on every SIMD, so the matrix pipe of a SIMD is fed by a wave that does
// nothing else and its vector ALU by another; one workgroup barrier per k-tile hands the buffers
// over.  The tile of each operand is three bf16 planes [x][k] (64-byte rows), the 16-byte chunk
// index XOR-ed with (x >> 2) & 3, which makes the fragment reads (one ds_read_b128 = 8
// consecutive k of a row) and the staging writes (ds_write_b64 = 4 consecutive k of a row, per
// plane) conflict-free for k-contiguous operands and 2-way for x-contiguous ones.  A producer
// thread owns 4 rows x 4 consecutive k of each operand:
//   [x][k] storage: row (t>>3) + 32a, k = 4(t&7) .. +3   (one 16-byte load, or four 4-byte loads)
//   [k][x] storage: row (t&31) + 32a, k = 4(t>>5) .. +3  (four 4-byte loads, 128 B per half-wave)
// A small B under many rows (a layer's weights: n*k <= 32M elements, m >= 2048) is split ONCE per
// call by gemm_split_b_kernel into the LDS images of its tiles (24 KB each, in the workspace);
// the producers then copy 96 bytes per thread and tile instead of splitting it again for each
// of the m/128 row tiles, and a k-major weight matrix (the input-gradient product) needs no
// 4-byte loads.
// Edges: out-of-range rows/columns read a clamped (valid) address and only ever reach
// accumulators that are not stored; the one partial k-tile is loaded through a masked path.
// Workgroup ids are remapped so that the workgroups sharing an XCD (ids equal mod 8) walk a
// contiguous range of tiles, n fastest: an A panel is then fetched into one L2 only.
// Bound: MFMA.  Algorithmic flops = 2*M*N*K; the matrix pipe executes six bf16 products per
// algorithmic one, so the peak of this path is the dense bf16 peak / 6 = 2500 / 6 = 417 TFLOP/s.
//
// A reduction over many rows into a small output (weight gradients: M,N ~ 1000, K = rows)
// is split over K into slabs and summed by a second kernel in slab order (deterministic).
#include "tv_common.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int STAGE_BYTES = BM * BN * 4;               // the output tile, staged in LDS (64 KB)
constexpr int STAGE_PIECES = 16;                       // ... and stored 4 rows per wave at a time

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
  int64_t units;             // work units of the launch: (tile, split) pairs, walked by a persistent grid
  int accumulate;
  int a_vec, b_vec;          // 16-byte loads allowed for the operand (host-side dispatch)
  int c_vec;                 // 16-byte stores allowed for the output (ldc, N, base all % 4)
  // operand prologue (tv_gemm_f32_act): v -> lrelu(v * act_scale[f] + act_shift[f]) on the way
  // into LDS, f = the operand's feature index (k for a row-major A, n for a k-major B)
  const float *act_scale, *act_shift;
  float act_slope;
  // B already split (gemm_split_b_kernel): LDS images of its tiles, [n-tile][k-tile][24 KB]
  const char *b_planes;
  int64_t b_ktiles;
  // the fp16 forms: 256 partial maxima of |operand| each (gemm_amax_kernel)
  const float *a_amax, *b_amax;
};

// Epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// The direct forms (edge tiles, tail slabs, row scale / accumulate); whole tiles whose rows can be
// written 16 bytes per lane go through the staged form below.
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, f32x16 (&acc)[2][2],
                                              int tile, int sp, bool tail, int nwg, int64_t m0,
                                              int64_t n0, int wave, int lane) {
  const int wm = wave >> 1, wn = wave & 1;
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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int PLANE_BYTES = BM * 64;            // one bf16 plane of one operand tile: 128 rows x 64 B
constexpr int OPERAND_BYTES = 3 * PLANE_BYTES;  // 24 KB
static_assert(BK == 32 && BM == BN, "plane layout assumes 32-deep tiles of 128 rows");

__device__ __forceinline__ unsigned plane_offset(int row, int piece) {   // piece = 8-byte unit 0..7
  return (unsigned)(row * 64 + ((((piece >> 1) ^ (row >> 2)) & 3) << 4) + ((piece & 1) << 3));
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {        // v_cvt_pk_bf16_f32
  const f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}


// The 16 fp32 elements a producer thread holds of one operand tile: element (row a, k j).
// They are the destinations of global loads issued with inline assembly, so that the loads of
// two tiles stay in flight across the loop: the compiler's own counting of outstanding loads
// merges the loop's paths and ends up waiting for the newest load before every split, which
// exposes the whole memory latency once per k-tile.  Nothing may read a set between its loads
// and the `landed()` that follows the matching s_waitcnt.
template <int MODE>
struct TileRegs {                                // 4-byte loads: one register per element
  float v[16];
  __device__ __forceinline__ float get(int a, int j) const { return v[4 * a + j]; }
  __device__ __forceinline__ void set(int a, int j, float x) { v[4 * a + j] = x; }
  __device__ __forceinline__ void landed() {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                      "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
  }
};
template <>
struct TileRegs<0> {                             // 16-byte loads: one register quad per row
  f32x4 q[4];
  __device__ __forceinline__ float get(int a, int j) const { return q[a][j]; }
  __device__ __forceinline__ void set(int a, int j, float x) { q[a][j] = x; }
  __device__ __forceinline__ void landed() {
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
  }
};

// ... plus, for an operand with a prologue along k, the scale and shift of the thread's four k
template <int MODE, bool COEF>
struct SetRegs : TileRegs<MODE> {};
template <int MODE>
struct SetRegs<MODE, true> : TileRegs<MODE> {
  f32x4 cs, ch;
  __device__ __forceinline__ void landed() {
    TileRegs<MODE>::landed();
    asm volatile("" : "+v"(cs), "+v"(ch));
  }
};

template <int N>
__device__ __forceinline__ void wait_loads_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N)); }

// MODE 0: [x][k] storage, 16-byte loads; 1: [x][k], 4-byte loads; 2: [k][x], 4-byte loads.
template <int MODE, bool ACT>
struct Split {
  // The coefficients of a prologue along k (MODE 0/1) travel with the tile, in the same register
  // set and by the same kind of load: a load the compiler counts would make it wait for every
  // load in flight, the two tiles of prefetch included.
  static constexpr bool COEF = ACT && MODE != 2;
  static constexpr int LOADS = (MODE == 0 ? 4 : 16) + (COEF ? 2 : 0);   // loads per tile and thread
  using Regs = SetRegs<MODE == 0 ? 0 : 1, COEF>;
  const char *tile;             // origin of the current k-tile (wave-uniform: stays in SGPRs)
  unsigned off[4];              // BYTE offset of row a's first k (MODE 2: of column a at the
                                // thread's first k) from `tile`
  unsigned kstep;               // MODE 2: ld in bytes
  int t, krow;                  // krow = first k of this thread inside the tile
  // prologue: lrelu(v * scale[f] + shift[f]); f = k for MODE 0/1 (A), f = x for MODE 2 (B)
  const float *act_scale, *act_shift;
  float slope;
  float sc[4], sh[4];           // MODE 2: fixed per thread
  unsigned kpos;                // MODE 0/1: the thread's first k in the tile the next load() fetches
  unsigned kfirst;              //           ... and in the first tile

  __device__ __forceinline__ int row(int a) const { return MODE == 2 ? (t & 31) + 32 * a : (t >> 3) + 32 * a; }
  __device__ __forceinline__ int piece() const { return MODE == 2 ? (t >> 5) : (t & 7); }

  __device__ __forceinline__ void init(const float *base, int64_t ld, int64_t x0, int64_t xlim,
                                       int64_t k0, int t_) {
    t = t_;
    krow = 4 * piece();
    tile = reinterpret_cast<const char *>(MODE == 2 ? base + k0 * ld + x0 : base + x0 * ld + k0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int64_t x = row(a);
      if (x0 + x >= xlim) x = xlim - 1 - x0;       // clamped rows only reach unstored accumulators
      off[a] = 4u * (MODE == 2 ? (unsigned)(krow * ld + x) : (unsigned)(x * ld + krow));
    }
    kstep = 4u * (unsigned)ld;
  }
  __device__ __forceinline__ void init_act(const float *scale, const float *shift, float slope_,
                                           int64_t x0, int64_t xlim, int64_t k0) {
    if (!ACT) return;
    act_scale = scale; act_shift = shift; slope = slope_;
    if (MODE == 2) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        int64_t x = x0 + row(a);
        if (x >= xlim) x = xlim - 1;
        sc[a] = scale[x]; sh[a] = shift[x];
      }
    } else {
      kpos = kfirst = (unsigned)(k0 + krow);
    }
  }
  __device__ __forceinline__ void advance() {
    tile += MODE == 2 ? (int64_t)BK * kstep : 4 * BK;
    if (COEF) kpos += BK;
  }

  __device__ __forceinline__ static void load16(f32x4 &dst, unsigned o, const char *base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(o), "s"(base));
  }
  __device__ __forceinline__ static void load4(float &dst, unsigned o, const char *base) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(o), "s"(base));
  }
  // issue the LOADS loads of a full k-tile
  __device__ __forceinline__ void load(Regs &r) const {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if constexpr (MODE == 0) {
        load16(r.q[a], off[a], tile);
      } else if constexpr (MODE == 1) {
        asm volatile("global_load_dword %0, %1, %2" : "=v"(r.v[4 * a]) : "v"(off[a]), "s"(tile));
        asm volatile("global_load_dword %0, %1, %2 offset:4" : "=v"(r.v[4 * a + 1]) : "v"(off[a]), "s"(tile));
        asm volatile("global_load_dword %0, %1, %2 offset:8" : "=v"(r.v[4 * a + 2]) : "v"(off[a]), "s"(tile));
        asm volatile("global_load_dword %0, %1, %2 offset:12" : "=v"(r.v[4 * a + 3]) : "v"(off[a]), "s"(tile));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) load4(r.v[4 * a + j], off[a] + j * kstep, tile);
      }
    }
    if constexpr (COEF) {
      load16(r.cs, 4u * kpos, reinterpret_cast<const char *>(act_scale));
      load16(r.ch, 4u * kpos, reinterpret_cast<const char *>(act_shift));
    }
  }
  // the same number of loads for the partial last k-tile: of its `rem` valid k; the others read
  // k 0 of the thread's row and are zeroed by zero_tail() once they have landed
  __device__ __forceinline__ void load_tail(Regs &r, int rem) const {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if constexpr (MODE == 0) {
        // rem % 4 == 0 here (16-byte loads need K % 4 == 0): the four k are valid together
        load16(r.q[a], krow < rem ? off[a] : off[a] - 4u * krow, tile);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = krow + j < rem;
          const unsigned o = MODE == 2 ? (ok ? off[a] + j * kstep : off[a] - krow * kstep)
                                       : (ok ? off[a] + 4u * j : off[a] - 4u * krow);
          load4(r.v[4 * a + j], o, tile);
        }
      }
    }
    if constexpr (COEF) {
      // the thread's four coefficients when they all exist, else the first four of the vectors
      // (K >= 4, checked on the host); zero_tail() puts the right values there
      const bool whole = krow + 4 <= rem;
      load16(r.cs, whole ? 4u * kpos : 0u, reinterpret_cast<const char *>(act_scale));
      load16(r.ch, whole ? 4u * kpos : 0u, reinterpret_cast<const char *>(act_shift));
    }
  }
  // `it` = index of the (partial) tile among the tiles of this workgroup
  __device__ __forceinline__ void zero_tail(Regs &r, int rem, int it) const {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (krow + j >= rem) r.set(a, j, 0.f);
    if constexpr (COEF) {
      if (krow + 4 > rem) {                      // coefficients of k that do not exist are zero
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = krow + j < rem;
          r.cs[j] = ok ? act_scale[kfirst + (unsigned)it * BK + j] : 0.f;
          r.ch[j] = ok ? act_shift[kfirst + (unsigned)it * BK + j] : 0.f;
        }
      }
    }
  }
  // the prologue on a landed tile.  Zero-filled k of a partial tile stay zero for f = k
  // (scale = shift = 0 there); for f = x they become lrelu(shift), harmless because the other
  // operand's zero-filled k multiply them.
  __device__ __forceinline__ void transform(Regs &r) const {
    if (!ACT) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float y;
        if constexpr (COEF) y = r.get(a, j) * r.cs[j] + r.ch[j];
        else y = r.get(a, j) * sc[a] + sh[a];
        r.set(a, j, fmaxf(y, y * slope));
      }
  }
  // three bf16 planes of the 16 elements into the operand's LDS tile
  __device__ __forceinline__ void split_store(const Regs &r, char *lds) const {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float x0 = r.get(a, 0), x1 = r.get(a, 1), x2 = r.get(a, 2), x3 = r.get(a, 3);
      const unsigned p01 = pack_bf16(x0, x1), p23 = pack_bf16(x2, x3);
      const float e0 = x0 - __uint_as_float(p01 << 16), e1 = x1 - __uint_as_float(p01 & 0xffff0000u);
      const float e2 = x2 - __uint_as_float(p23 << 16), e3 = x3 - __uint_as_float(p23 & 0xffff0000u);
      const unsigned q01 = pack_bf16(e0, e1), q23 = pack_bf16(e2, e3);
      const float f0 = e0 - __uint_as_float(q01 << 16), f1 = e1 - __uint_as_float(q01 & 0xffff0000u);
      const float f2 = e2 - __uint_as_float(q23 << 16), f3 = e3 - __uint_as_float(q23 & 0xffff0000u);
      const unsigned s01 = pack_bf16(f0, f1), s23 = pack_bf16(f2, f3);
      char *dst = lds + plane_offset(row(a), piece());
      *reinterpret_cast<uint2 *>(dst) = make_uint2(p01, p23);
      *reinterpret_cast<uint2 *>(dst + PLANE_BYTES) = make_uint2(q01, q23);
      *reinterpret_cast<uint2 *>(dst + 2 * PLANE_BYTES) = make_uint2(s01, s23);
    }
  }
};

// An operand that gemm_split_b_kernel has already split: its tiles lie in memory as the LDS
// image (three planes, swizzled), 24 KB each, and a producer thread copies 6 x 16 bytes of it.
// Same interface as Split, so the producer loop does not care which it has.
struct PlaneTile {
  static constexpr int LOADS = 6;
  struct Regs {
    f32x4 q[6];
    __device__ __forceinline__ void landed() {
      asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]));
    }
  };
  const char *tile;
  unsigned off;                 // 16 t: the thread's first 16 bytes of the image
  int krow;
  __device__ __forceinline__ void init(const char *planes, int64_t ktiles, int64_t tile_n, int64_t k0, int t) {
    tile = planes + (tile_n * ktiles + k0 / BK) * OPERAND_BYTES;
    off = 16u * (unsigned)t;
    krow = 0;
  }
  __device__ __forceinline__ void init_act(const float *, const float *, float, int64_t, int64_t, int64_t) {}
  __device__ __forceinline__ void advance() { tile += OPERAND_BYTES; }
  __device__ __forceinline__ void load(Regs &r) const {
#pragma unroll
    for (int j = 0; j < 6; ++j)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.q[j]) : "v"(off + 4096u * j), "s"(tile));
  }
  __device__ __forceinline__ void load_tail(Regs &r, int) const { load(r); }   // images are zero-padded in k
  __device__ __forceinline__ void zero_tail(Regs &, int, int) const {}
  __device__ __forceinline__ void transform(Regs &) const {}
  __device__ __forceinline__ void split_store(const Regs &r, char *lds) const {
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4 *>(lds + off + 4096u * j) = r.q[j];
  }
};

// B (the small operand of a layer product: the weight matrix) split once per call instead of
// once per output tile: one workgroup writes the 24 KB LDS image of one (n-tile, k-tile) pair,
// zero outside the matrix.  kx: B is stored [k][n] (trans_b = 0), else [n][k].
__device__ __forceinline__ void gemm_split_b_block(int64_t block, const float *B, int64_t ldb, int kx,
                                                   int64_t N, int64_t K, int64_t ktiles, char *planes) {
  const int t = threadIdx.x;
  const int64_t tn = block / ktiles, kt = block % ktiles;
  char *image = planes + block * OPERAND_BYTES;
  // thread -> (4 rows, 4 consecutive k) as the in-kernel split, following the storage order
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = kx ? (t & 31) + 32 * a : (t >> 3) + 32 * a;
    const int piece = kx ? (t >> 5) : (t & 7);
    const int64_t n = tn * BN + row, k0 = kt * BK + 4 * piece;
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t k = k0 + j;
      x[j] = (n < N && k < K) ? (kx ? B[k * ldb + n] : B[n * ldb + k]) : 0.f;
    }
    const unsigned p01 = pack_bf16(x[0], x[1]), p23 = pack_bf16(x[2], x[3]);
    const float e0 = x[0] - __uint_as_float(p01 << 16), e1 = x[1] - __uint_as_float(p01 & 0xffff0000u);
    const float e2 = x[2] - __uint_as_float(p23 << 16), e3 = x[3] - __uint_as_float(p23 & 0xffff0000u);
    const unsigned q01 = pack_bf16(e0, e1), q23 = pack_bf16(e2, e3);
    const float f0 = e0 - __uint_as_float(q01 << 16), f1 = e1 - __uint_as_float(q01 & 0xffff0000u);
    const float f2 = e2 - __uint_as_float(q23 << 16), f3 = e3 - __uint_as_float(q23 & 0xffff0000u);
    const unsigned s01 = pack_bf16(f0, f1), s23 = pack_bf16(f2, f3);
    char *dst = image + plane_offset(row, piece);
    *reinterpret_cast<uint2 *>(dst) = make_uint2(p01, p23);
    *reinterpret_cast<uint2 *>(dst + PLANE_BYTES) = make_uint2(q01, q23);
    *reinterpret_cast<uint2 *>(dst + 2 * PLANE_BYTES) = make_uint2(s01, s23);
  }
}

// the three planes of one 32-row sub-tile for one 16-deep k-step
__device__ __forceinline__ void read_set(const char *lds, int row0, int lane, int step, bf16x8 (&f)[3]) {
  const int rowi = row0 + (lane & 31);
  const char *p = lds + plane_offset(rowi, 2 * (2 * step + (lane >> 5)));
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8 *>(p + pl * PLANE_BYTES);
}

// the six partial products of one 32x32 tile for one k-step: a0 b0 -> hi, the corrections -> lo
// (six back-to-back matrix instructions, nothing for the vector ALU to do)
__device__ __forceinline__ void mma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 &hi, f32x16 &lo) {
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
}

// Staged epilogue.  A finished 64x64 block goes into the wave's own 16 KB of LDS behind the
// plane buffers, transposed to row-major, and leaves it in 16 pieces of 4 rows (one 16-byte store
// per lane each) that the wave issues one per k-tile DURING THE NEXT UNIT's matrix instructions.
// Written out at once, the 64 KB tiles of all 256 CUs arrive at the memory system together and
// the producers' loads for the next unit queue behind them: measured 10-16 % of the cycles of a
// 51200 x 1000 x 1000 product (the same kernel with the stores removed).  The region belongs to
// one wave and a wave's LDS accesses execute in order, so no barrier is involved.  Rows are 64
// floats (all 64 banks); the two 32-lane halves of an accumulator register sit 4 rows apart and
// swap 32-column halves (col ^ 32 when (row >> 2) & 1) to stay on different banks.
struct Pending {
  float *o;            // row (lane >> 4) of the block, the lane's 4 columns; nullptr columns: colok
  float4 bv;           // bias of those columns
  int64_t ldo;
  bool colok;
  bool stream;         // final output (streams out) or split-K slab (read back at once)
  int left;            // pieces not yet stored (0: nothing pending); piece q covers rows 4q .. 4q+3
};

__device__ __forceinline__ void stage_block(float *ws, f32x16 (&acc)[2][2], int lane) {
  const int kh = lane >> 5, xl = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ws[(i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2)) * 64 + ((j * 32 + xl) ^ (kh << 5))] = acc[i][j][r];
}
__device__ __forceinline__ float4 staged_piece(const float *ws, int q, int lane) {
  return *reinterpret_cast<const float4 *>(&ws[(q * 4 + (lane >> 4)) * 64 + (((lane & 15) * 4) ^ ((q & 1) << 5))]);
}
__device__ __forceinline__ void store_piece(const Pending &p, int q, const float4 &v) {
  if (p.colok) {
    const f32x4 o = {v.x + p.bv.x, v.y + p.bv.y, v.z + p.bv.z, v.w + p.bv.w};
    f32x4 *dst = reinterpret_cast<f32x4 *>(p.o + (int64_t)q * 4 * p.ldo);
    // the output streams out (non-temporal: it is far larger than the L2 and would push the
    // operand panels out of it; +4 % on a 51200 x 1000 x 1000 product); split-K slabs are read
    // back by the reduction kernel straight away and stay cacheable
    if (p.stream) __builtin_nontemporal_store(o, dst);
    else *dst = o;
  }
}

// One work unit of a launch: an output tile, or one k-split of it.
struct Unit {
  int tile, sp, nwg;
  bool tail;
  int tn;
  int64_t m0, n0, kbeg;
  int ntf, rem, nt;            // full k-tiles, k of the partial one, k-tiles in all
};

// unit id -> (output tile, k-split).  Unit ids equal mod 8 walk a contiguous range of tiles (n
// fastest): a persistent workgroup w takes units w, w + G, w + 2G, ... with G a multiple of 8, so
// the workgroups of one XCD (hardware ids equal mod 8) share A panels in their own L2.
__device__ __forceinline__ Unit decode_unit(const GemmArgs &g, int u) {
  Unit x;
  const int tail_units = g.tail_splits > 1;
  const int nwg = tail_units ? (int)g.tail_first : (int)g.units;
  int tile, sp;
  int64_t kchunk;
  x.tail = tail_units && u >= nwg;
  if (x.tail) {
    const int v = u - nwg;
    tile = nwg + v / (int)g.tail_splits;
    sp = v % (int)g.tail_splits;
    kchunk = g.tail_kchunk;
  } else {
    const int q = nwg >> 3, rr = nwg & 7;
    const int xcd = u & 7, idx = u >> 3;
    const int L = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    const int tiles = (int)(g.tiles_n * g.tiles_m);
    tile = L % tiles;
    sp = L / tiles;
    kchunk = g.k_chunk;
  }
  x.nwg = nwg;
  x.tile = __builtin_amdgcn_readfirstlane(tile);
  x.sp = __builtin_amdgcn_readfirstlane(sp);
  x.tn = __builtin_amdgcn_readfirstlane(x.tile % (int)g.tiles_n);
  const int tm = __builtin_amdgcn_readfirstlane(x.tile / (int)g.tiles_n);
  x.m0 = (int64_t)tm * BM;
  x.n0 = (int64_t)x.tn * BN;
  x.kbeg = x.sp * kchunk;
  const int64_t kend = (x.kbeg + kchunk < g.K) ? (x.kbeg + kchunk) : g.K;
  const int64_t klen = kend > x.kbeg ? kend - x.kbeg : 0;
  x.ntf = __builtin_amdgcn_readfirstlane((int)(klen / BK));
  x.rem = __builtin_amdgcn_readfirstlane((int)(klen - (int64_t)x.ntf * BK));
  x.nt = x.ntf + (x.rem ? 1 : 0);
  return x;
}

// 512 threads = four consumer waves (0-3) and four producer waves (4-7): see the head of the file.
// The grid is PERSISTENT: one workgroup per CU walks the units of the launch, so while the
// consumers add, transpose and store the tile they have finished (in an LDS region of their
// own), the producers already load and split the first k-tiles of the next unit; the stores
// drain behind the next unit's matrix instructions.  (With one workgroup per unit the epilogue
// and the next workgroup's first loads were exposed on every CU at once, 14-20 % of a
// 1000-deep product.)
// Measured per k-tile at 51200 x 1000 x 1000 (in-kernel cycle stamps): consumers 1875 cycles
// between barriers (1536 = 48 matrix instructions of 32 cycles), producers 2050-2230 for about 210
// vector instructions, 12 LDS stores and 8 loads beside them - the vector side is the longer
// pole of a tile.  (Earlier forms - waves that alternate split and multiply phases, alone or as
// two workgroups in anti-phase - left the matrix pipe 44 % busy: the two phases did not overlap.)
template <int A_MODE, int B_MODE, int ACT>
__global__ __launch_bounds__(512) void gemm_kernel(GemmArgs g) {
  static_assert(ACT == 0 || (ACT == 1 && A_MODE != 2) || (ACT == 2 && B_MODE == 2), "prologue orientation");
  static_assert(B_MODE >= 0 && B_MODE <= 3 && A_MODE >= 0 && A_MODE <= 2, "load modes");
  extern __shared__ __align__(16) char smem[];     // [buffer][A planes | B planes] | epilogue staging
  const bool producer = threadIdx.x >= 256;
  const int t = threadIdx.x & 255;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int units = (int)g.units;

  if (producer) {
    using OperandA = Split<A_MODE, ACT == 1>;
    using OperandB = typename std::conditional<B_MODE == 3, PlaneTile, Split<B_MODE == 3 ? 0 : B_MODE, ACT == 2>>::type;
    constexpr int TILE_LOADS = OperandA::LOADS + OperandB::LOADS;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
      const Unit x = decode_unit(g, u);
      const int ntf = x.ntf, rem = x.rem, nt = x.nt;
      OperandA sa;
      OperandB sb;
      sa.init(g.A, g.lda, x.m0, g.M, x.kbeg, t);
      if constexpr (B_MODE == 3) sb.init(g.b_planes, g.b_ktiles, x.tn, x.kbeg, t);
      else sb.init(g.B, g.ldb, x.n0, g.N, x.kbeg, t);
      sa.init_act(g.act_scale, g.act_shift, g.act_slope, x.m0, g.M, x.kbeg);
      sb.init_act(g.act_scale, g.act_shift, g.act_slope, x.n0, g.N, x.kbeg);
      // Register sets: tile `it` lives in set it % DEPTH and the loads of tile it+DEPTH are issued
      // as soon as tile it is split, so when tile it is needed only the loads of the (at most
      // DEPTH-1) tiles after it are younger than its own.  Three tiles in flight: the latency of
      // a load has a long tail while the memory system also takes the output tiles, and two
      // k-tiles (2.5 us) of lookahead did not cover it.
      constexpr int DEPTH = 3;
      constexpr int YOUNGER2 = 2 * TILE_LOADS < 63 ? 2 * TILE_LOADS : 63;   // vmcnt holds 6 bits
      typename OperandA::Regs ra[DEPTH];
      typename OperandB::Regs rb[DEPTH];
#define TV_LOAD(IT, RA, RB)                                                       \
  do {                                                                            \
    if ((IT) < ntf) { sa.load(RA); sb.load(RB); }                                 \
    else if ((IT) < nt) { sa.load_tail(RA, rem); sb.load_tail(RB, rem); }         \
    if ((IT) < nt) { sa.advance(); sb.advance(); }                                \
  } while (0)
#define TV_STAGE(IT, RA, RB)                                                      \
  do {                                                                            \
    if ((IT) < nt) {                                                              \
      char *buf = smem + ((IT) & 1) * (2 * OPERAND_BYTES);                        \
      if ((IT) + 2 < nt) wait_loads_but<YOUNGER2>();                              \
      else if ((IT) + 1 < nt) wait_loads_but<TILE_LOADS>();                       \
      else wait_loads_but<0>();                                                   \
      RA.landed(); RB.landed();                                                   \
      if ((IT) >= ntf) { sa.zero_tail(RA, rem, (IT)); sb.zero_tail(RB, rem, (IT)); } \
      sa.transform(RA); sb.transform(RB);                                         \
      sa.split_store(RA, buf);                                                    \
      sb.split_store(RB, buf + OPERAND_BYTES);                                    \
      TV_LOAD((IT) + DEPTH, RA, RB);                                              \
    }                                                                             \
    __syncthreads();                                                              \
  } while (0)
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) TV_LOAD(d, ra[d], rb[d]);
      // nt + 1 barriers per unit, like the consumers.  The last one follows the consumers' last
      // read of a plane buffer, so the next unit's tile 0 may be written straight after it.
      for (int it = 0; it <= nt; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
          if (it + d <= nt) TV_STAGE(it + d, ra[d], rb[d]);
      }
#undef TV_STAGE
#undef TV_LOAD
    }
    return;
  }

  float *ws = reinterpret_cast<float *>(smem + 4 * OPERAND_BYTES) + wave * (64 * 64);
  Pending pend;
  pend.o = nullptr; pend.bv = make_float4(0.f, 0.f, 0.f, 0.f); pend.ldo = 0; pend.colok = false; pend.stream = false; pend.left = 0;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const Unit x = decode_unit(g, u);
    const int rem = x.rem, nt = x.nt;
    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }

    __syncthreads();                               // tile 0 is in buffer 0
    // a partial last tile of at most 16 k holds nothing in its second k-step (K = 1000: 8 of 32)
    const int nfull = (rem > 0 && rem <= 16) ? nt - 1 : nt;
    for (int it = 0; it < nfull; ++it) {
      const char *As = smem + (it & 1) * (2 * OPERAND_BYTES), *Bs = As + OPERAND_BYTES;
      // Both 16-deep k-steps (a partial tile's missing k were stored as zeros).  Four fragment
      // sets (A0, B0, B1, A1 of a step; 3 planes = 12 registers each) are refilled as soon as
      // their last use has been issued, one or two stages (6-12 MFMAs, 200-400 cycles) ahead of
      // their next use; the stage order is pinned, so the matrix pipe only waits for LDS at the
      // head of a tile.
      //   stage : 1      2      3      4    | 5       6      7      8
      //   tile  : (0,0)  (0,1)  (1,1)  (1,0)| (0,1)   (0,0)  (1,0)  (1,1)
      //   sets  : P Q    P R    S R    S Q  | P R     P Q    S Q    S R
      //   refill: -      -      P=A0'  R=B1'| Q=B0',S=A1'
      // Piece `it` of the previous unit's staged block is read with the first fragments and
      // stored after stage 4.
      const bool emit = it < pend.left;            // pieces go out in order: piece it at k-tile it
      bf16x8 P[3], Q[3], R[3], S[3];
      float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
      read_set(As, wm * 64, lane, 0, P);           // A0 step 0
      read_set(Bs, wn * 64, lane, 0, Q);           // B0 step 0
      read_set(Bs, wn * 64 + 32, lane, 0, R);      // B1 step 0
      read_set(As, wm * 64 + 32, lane, 0, S);      // A1 step 0
      if (emit) pv = staged_piece(ws, it, lane);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, Q, hi[0][0], lo[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, R, hi[0][1], lo[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      read_set(As, wm * 64, lane, 1, P);           // A0 step 1
      mma6(S, R, hi[1][1], lo[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      read_set(Bs, wn * 64 + 32, lane, 1, R);      // B1 step 1
      mma6(S, Q, hi[1][0], lo[1][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (emit) store_piece(pend, it, pv);
      read_set(Bs, wn * 64, lane, 1, Q);           // B0 step 1
      read_set(As, wm * 64 + 32, lane, 1, S);      // A1 step 1
      mma6(P, R, hi[0][1], lo[0][1]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(P, Q, hi[0][0], lo[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(S, Q, hi[1][0], lo[1][0]);
      __builtin_amdgcn_sched_barrier(0);
      mma6(S, R, hi[1][1], lo[1][1]);
      __syncthreads();                             // buffer it&1 may be refilled, (it+1)&1 is ready
    }
    if (nfull < nt) {                              // first k-step only
      const char *As = smem + (nfull & 1) * (2 * OPERAND_BYTES), *Bs = As + OPERAND_BYTES;
      bf16x8 P[3], Q[3], R[3], S[3];
      read_set(As, wm * 64, lane, 0, P);
      read_set(Bs, wn * 64, lane, 0, Q);
      read_set(Bs, wn * 64 + 32, lane, 0, R);
      read_set(As, wm * 64 + 32, lane, 0, S);
      mma6(P, Q, hi[0][0], lo[0][0]);
      mma6(P, R, hi[0][1], lo[0][1]);
      mma6(S, R, hi[1][1], lo[1][1]);
      mma6(S, Q, hi[1][0], lo[1][0]);
      __syncthreads();
    }
    // a unit of fewer than 16 k-tiles leaves pieces of the previous block behind
    for (int q = nfull; q < pend.left; ++q) store_piece(pend, q, staged_piece(ws, q, lane));
    pend.left = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) hi[i][j][r] += lo[i][j][r];

    // every consumer is past the unit's last barrier; the plane buffers now belong to the
    // producers again (next unit), the staging region behind them to this wave
    const bool split = g.splits > 1;
    const bool plain = split || (!g.row_scale && !g.accumulate);
    if (!x.tail && plain && g.c_vec && x.m0 + BM <= g.M) {
      // all 128 rows of the tile exist and rows can be written 16 bytes per lane (N % 4 == 0: a
      // lane's four columns exist together)
      stage_block(ws, hi, lane);
      const int64_t col = x.n0 + wn * 64 + (lane & 15) * 4;
      float *out = split ? (g.slabs + (int64_t)x.sp * g.M * g.N) : g.C;
      pend.ldo = split ? g.N : g.ldc;
      pend.colok = col < g.N;
      pend.stream = !split;
      pend.o = out + (x.m0 + wm * 64 + (lane >> 4)) * pend.ldo + col;
      pend.bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pend.colok && !split && g.bias)
        pend.bv = make_float4(g.bias[col], g.bias[col + 1], g.bias[col + 2], g.bias[col + 3]);
      pend.left = STAGE_PIECES;
    } else {
      gemm_epilogue(g, hi, x.tile, x.sp, x.tail, x.nwg, x.m0, x.n0, wave, lane);
    }
  }
  for (int q = 0; q < pend.left; ++q) store_piece(pend, q, staged_piece(ws, q, lane));
}

// ------------------------------------------------------------------------------------------
// The wide form: the layer products that carry the time (many rows of a row-major A against a
// small pre-split B: forward products and input gradients).  Its kernel, gemm_wide_h_kernel, is
// further down with the fp16 arithmetic it runs on; the shape of it:
//
// 128 x 256 output tile, 512 threads = EIGHT waves that all multiply (2 x 4, a 64x64 block each),
// two per SIMD: while one wave waits for its fragments or at the barrier the other keeps the
// matrix pipe busy.  B's two plane images come global -> LDS by LDS-DMA (global_load_lds_dwordx4:
// no registers, no ds_write, no vector ALU), A is split by the multiplying waves themselves
// between their matrix instructions: 8 elements per thread and k-tile (two 16-byte loads, the
// split, two ds_write_b128).  Two LDS buffers, one barrier per k-tile, loads and DMAs counted by
// hand (inline assembly: beside LDS-DMA in flight the compiler drains every load before a
// ds_write of its own).  tools/gemm_planes_lab.hip measured the form first, on three bf16 planes
// (200 TF on 51200 x 1024 x 1024 against 177 for four multiplying waves fed by four producer
// waves); tools/gemm_f16x2_lab.hip the same form on two fp16 planes (320 TF).
// Work units: output tiles, plus the tail split of the last partial round (packed 128x256 slabs,
// gemm_wide_tail_reduce_kernel).
constexpr int WBN = 256;
// second stage of the wide form's tail split: C tile = sum_s slab[tile][s] + bias in split order.
// One workgroup per 16 rows of a tail tile; a thread owns 4 consecutive columns of 4 rows.
__global__ __launch_bounds__(256) void gemm_wide_tail_reduce_kernel(GemmArgs g) {
  const int64_t rel = blockIdx.x >> 3;
  const int64_t tile = g.tail_first + rel;
  const int64_t tn = tile % g.tiles_n, tm = tile / g.tiles_n;
  const int t = threadIdx.x;
  const int c4 = (t & 63) * 4;
  const float *slab = g.tail_slabs + rel * g.tail_splits * (BM * WBN);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = (blockIdx.x & 7) * 16 + (t >> 6) + 4 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t s = 0; s < g.tail_splits; ++s) {
      const float4 p = *reinterpret_cast<const float4 *>(&slab[s * (BM * WBN) + rl * WBN + c4]);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    const int64_t row = tm * BM + rl;
    if (row >= g.M) continue;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = tn * WBN + c4 + j;
      if (col < g.N) g.C[row * g.ldc + col] = e[j] + (g.bias ? g.bias[col] : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The wide form for k-major operands: the weight gradients, C[M,N] = A^T B with A = [K][M] (a
// layer's output gradient) and B = [K][N] (its input, optionally behind the deferred
// BatchNorm + LeakyReLU prologue), K = graph nodes of the batch, M and N = layer widths.
//
// (gemm_kmajor_wide_h_kernel, further down.)  Same shape as the wide form (128 x 256 tile, eight
// multiplying waves, two LDS buffers, one barrier per k-tile), but a thread that loads 16 bytes of
// such an operand holds ONE k of four x where a matrix-instruction operand wants eight k of one
// x.  The planes therefore stay k-major in LDS ([k][x] rows of 256 bytes of 16-bit elements, the
// 16-byte chunk index XOR-ed with the row: the dual-use image of the MI355X guide), a thread
// stores its four elements of a plane with one ds_write_b64, and the fragments come out through
// the transposing read of gfx950
// (ds_read_b64_tr_b16: a 4-row x 16-column block per 16 lanes, delivered column-major; two of
// them make the eight k of a lane).  Per thread and k-tile: six 16-byte loads (two of A, four of
// B) where the producer / consumer kernel needs 32 4-byte loads per producer thread; each float4
// is split after one of the six first blocks of matrix instructions of the tile before, so that
// its vector instructions issue between the dependent matrix instructions of the block, and
// its registers take the load of the tile after next at once (one register set, the count of
// loads in flight is the same at every wait: 5 younger ones).
// tools/gemm_kmajor_lab.hip measured the form first, on three bf16 planes: 196 TF on
// 1024 x 1024 x 51200 in 256 units against 163-167 for the producer / consumer kernel.
// Work units: (tile, k-split) pairs, one round of the chip, k-split fastest so that the units of
// one k-range (which share its rows of A and B) run on one XCD; packed 128 x 256 slabs, summed in
// split order by gemm_wide_tail_reduce_kernel.  M % 4 == 0, and N % 4 == 0 or rows of B padded
// to a multiple of 4 (a float4 of an operand row is inside or outside as a whole, or straddles N
// over readable padding; outside ones read column 0 of the tile; either way the extra columns
// only reach unstored accumulators).  The rows of a partial last k-tile are multiplied after the pipelined loop, in
// a step of their own whose loads are clamped to row K - 1 and whose A elements beyond K are zero.
__device__ __forceinline__ unsigned kmajor_offset(int k, int ch) {   // 16-byte chunk ch of row k
  return (unsigned)(256 * k + 16 * (ch ^ (((k & 3) << 2) | ((k >> 2) & 3))));
}

// ------------------------------------------------------------------------------------------
// The two wide forms on TWO fp16 planes per operand and THREE matrix instructions per product.
//
//   x s = x0 + 2^-11 x1,   x0 = fp16(x s),   x1 = fp16((x s - x0) 2^11)        (11 + 11 bits)
//   C   = [a0 b0] + 2^-11 [a0 b1 + a1 b0],   the brackets in separate fp32 accumulators
//
// s is a power of two per OPERAND (exact to apply and to take out again) that puts max|x| into
// [2^14, 2^15): fp16 overflows at 2^16, elements down to 2^-29 of the operand's maximum keep all
// 22 bits and smaller ones lose one bit per binade - an absolute error below 2^-37 of the maximum.
// The residual is kept at the scale of x0 (Ootomo & Yokota, "Recovering single precision accuracy
// from Tensor Cores ..."), so it cannot underflow before x0 does.  fp16 x fp16 products are exact
// in fp32; the dropped a1 b1 and the rounding of x1 are below 2^-22 |a b| each.  Measured against
// float64 (tools/gemm_f16x2_lab.hip, 51200 x 1024 x 1024 with row magnitudes over 20 binades):
// 6.0e-8 of sum|a||b| against 5.3e-8 for the three-plane bf16 form and 1.6e-7 for an fp32 fma
// chain - and 320 TFLOP/s against 200 on the same box: half the matrix instructions and two
// thirds of the LDS bytes per k-tile, on a chip whose clock follows the energy of a k-tile.
// max|x| comes from gemm_amax_kernel (a pass over the operand, behind the same prologue as the
// product applies; 256 partial maxima that every workgroup of the product reduces itself: no
// atomics, no host round trip).  Operands that hold inf / NaN give NaN.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int H_OPERAND_BYTES = 2 * PLANE_BYTES;       // 16 KB
constexpr int H_BUF_BYTES = 3 * H_OPERAND_BYTES;       // A image | two B images
constexpr int AMAX_SLOTS = 256;
constexpr float H_LO = 1.0f / 2048.0f;

__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  const f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// two elements -> their packed head and (scaled) residual halves
// (the head saturates at the largest fp16 instead of overflowing: an operand maximum handed in by
// the caller may be a bound that rows outside its statistics exceed, and a finite wrong value in
// a row nobody reads is harmless where an infinity is not: inf - inf = NaN in the residual)
__device__ __forceinline__ void split_f16(float a, float b, unsigned &head, unsigned &res) {
  a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
  b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
  head = pack_f16(a, b);
  const f32x2v h = __builtin_convertvector(__builtin_bit_cast(f16x2, head), f32x2v);
  res = pack_f16((a - h[0]) * 2048.f, (b - h[1]) * 2048.f);
}
// power-of-two scale that puts amax into [2^14, 2^15), and its inverse
__device__ __forceinline__ void pow2_scale(float amax, float &s, float &inv) {
  const unsigned b = (__float_as_uint(amax) >> 23) & 255u;
  if (b < 15u || b == 255u) { s = 1.f; inv = 1.f; return; }   // zero / denormal-range / inf, NaN
  s = __uint_as_float((268u - b) << 23);
  inv = __uint_as_float((b - 14u) << 23);
}
__device__ __forceinline__ float reduce_amax_slots(const float *slots, int lane) {
  float m = fmaxf(fmaxf(slots[lane], slots[lane + 64]), fmaxf(slots[lane + 128], slots[lane + 192]));
  m = tv_wave_max(m);
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}

// max |v| over a row-major [rows][cols] operand, v = x or lrelu(x * scale[col] + shift[col]).
// 256 workgroups of 1024 threads; thread t walks the float4 columns t % 256, t % 256 + 256, ... of
// rows ((it * 256 + wg) * 4 + t / 256) * 4 + u, u = 0..3 (sixteen rows of loads in flight per
// workgroup); columns beyond a multiple of four, or an operand that cannot be read 16 bytes at a
// time, go one float at a time.  A NaN is skipped by fmaxf and reaches the result through the
// product itself.
template <bool ACT>
__global__ __launch_bounds__(1024) void gemm_amax_kernel(const float *__restrict__ X, int64_t ld,
                                                         int64_t rows, int64_t cols, int vec,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ shift, float slope,
                                                         float *__restrict__ slots) {
  __shared__ float s_red[16];
  const int t = threadIdx.x, sub = t >> 8, tc = t & 255;
  const int64_t c4n = vec ? cols >> 2 : 0;
  float m = 0.f;
  for (int64_t c4 = tc; c4 < c4n; c4 += 256) {
    f32x4 cs = {1.f, 1.f, 1.f, 1.f}, ch = {0.f, 0.f, 0.f, 0.f};
    if constexpr (ACT) {
      cs = *reinterpret_cast<const f32x4 *>(scale + 4 * c4);
      ch = *reinterpret_cast<const f32x4 *>(shift + 4 * c4);
    }
    for (int64_t r0 = (((int64_t)blockIdx.x) * 4 + sub) * 4; r0 < rows; r0 += 256 * 16) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + u < rows ? r0 + u : rows - 1;
        v[u] = *reinterpret_cast<const f32x4 *>(X + r * ld + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = v[u][j];
          if constexpr (ACT) { a = a * cs[j] + ch[j]; a = fmaxf(a, a * slope); }
          m = fmaxf(m, fabsf(a));
        }
    }
  }
  // what is left: columns 4 c4n .. cols - 1
  const int64_t cfirst = 4 * c4n;
  for (int64_t c = cfirst + tc; c < cols; c += 256) {
    const float cs = ACT ? scale[c] : 1.f, ch = ACT ? shift[c] : 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + sub; r < rows; r += 256 * 4) {
      float a = X[r * ld + c];
      if constexpr (ACT) { a = a * cs + ch; a = fmaxf(a, a * slope); }
      m = fmaxf(m, fabsf(a));
    }
  }
  m = tv_wave_max(m);
  if ((t & 63) == 0) s_red[t >> 6] = m;
  __syncthreads();
  if (t == 0) {
    float w = s_red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) w = fmaxf(w, s_red[i]);
    slots[blockIdx.x] = w;
  }
}

// B split once per call into the fp16 LDS images of its tiles (16 KB each), scaled by the power
// of two that `slots` (its gemm_amax_kernel result) gives.  Layout of a plane as in the bf16 form.
__global__ __launch_bounds__(256) void gemm_split_b_kernel(const float *B, int64_t ldb, int kx,
                                                           int64_t N, int64_t K, int64_t ktiles,
                                                           char *planes) {
  gemm_split_b_block(blockIdx.x, B, ldb, kx, N, K, ktiles, planes);
}
__device__ __forceinline__ void gemm_split_b_h_block(int64_t block, const float *B, int64_t ldb,
                                                     int kx, int64_t N, int64_t K, int64_t ktiles,
                                                     const float *slots, char *planes) {
  const int t = threadIdx.x;
  float sb, inv;
  pow2_scale(reduce_amax_slots(slots, t & 63), sb, inv);
  const int64_t tn = block / ktiles, kt = block % ktiles;
  char *image = planes + block * H_OPERAND_BYTES;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = kx ? (t & 31) + 32 * a : (t >> 3) + 32 * a;
    const int piece = kx ? (t >> 5) : (t & 7);
    const int64_t n = tn * BN + row, k0 = kt * BK + 4 * piece;
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t k = k0 + j;
      x[j] = (n < N && k < K) ? (kx ? B[k * ldb + n] : B[n * ldb + k]) * sb : 0.f;
    }
    unsigned p01, p23, q01, q23;
    split_f16(x[0], x[1], p01, q01);
    split_f16(x[2], x[3], p23, q23);
    char *dst = image + plane_offset(row, piece);
    *reinterpret_cast<uint2 *>(dst) = make_uint2(p01, p23);
    *reinterpret_cast<uint2 *>(dst + PLANE_BYTES) = make_uint2(q01, q23);
  }
}
__global__ __launch_bounds__(256) void gemm_split_b_h_kernel(const float *B, int64_t ldb, int kx,
                                                             int64_t N, int64_t K, int64_t ktiles,
                                                             const float *slots, char *planes) {
  gemm_split_b_h_block(blockIdx.x, B, ldb, kx, N, K, ktiles, slots, planes);
}

// Several small operands split in ONE launch (tv_gemm_prepare_b_batch: the weights of a stack of
// layers, each in the two orientations its forward product and its input gradient read):
// blockIdx.y picks the operand, blockIdx.x its (column tile, k-tile) image.
constexpr int SPLIT_BATCH = 32;
struct SplitItem {
  const float *b;
  const float *slots;     // fp16 form: the operand's partial maxima
  char *planes;
  int64_t ldb, n, k, ktiles, blocks;
  int kx, form;
};
struct SplitBatch { SplitItem item[SPLIT_BATCH]; };
__global__ __launch_bounds__(256) void gemm_split_b_batch_kernel(SplitBatch batch) {
  const SplitItem &it = batch.item[blockIdx.y];
  if ((int64_t)blockIdx.x >= it.blocks) return;
  if (it.form) gemm_split_b_h_block(blockIdx.x, it.b, it.ldb, it.kx, it.n, it.k, it.ktiles, it.slots, it.planes);
  else gemm_split_b_block(blockIdx.x, it.b, it.ldb, it.kx, it.n, it.k, it.ktiles, it.planes);
}

__device__ __forceinline__ void read_set_h(const char *lds, int row0, int lane, int step, f16x8 (&f)[2]) {
  const int rowi = row0 + (lane & 31);
  const char *p = lds + plane_offset(rowi, 2 * (2 * step + (lane >> 5)));
  f[0] = *reinterpret_cast<const f16x8 *>(p);
  f[1] = *reinterpret_cast<const f16x8 *>(p + PLANE_BYTES);
}
// the three partial products of one 32x32 tile for one k-step: a0 b0 -> hi, the corrections -> lo
__device__ __forceinline__ void mma3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x16 &hi, f32x16 &lo) {
  lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], hi, 0, 0, 0);
}

__device__ __forceinline__ void wide_stage_b_h(const char *b_img, int64_t b_stride, char *buf, int wave, int lane) {
  typedef __attribute__((address_space(3))) void lds_void;
  // 32 chunks of 1 KB over 8 waves: 2 per image and wave, lane-linear (the images are byte copies)
  const unsigned voff = (unsigned)(wave * 2 * 1024 + lane * 16), loff = (unsigned)(wave * 2 * 1024);
#pragma unroll
  for (int img = 0; img < 2; ++img)
#pragma unroll
    for (int c = 0; c < 2; ++c)
      __builtin_amdgcn_global_load_lds((const void *)(b_img + (int64_t)img * b_stride + voff + c * 1024),
                                       (lds_void *)(buf + img * H_OPERAND_BYTES + loff + c * 1024), 16, 0, 0);
}

// two fp16 planes of 8 consecutive k of one row into the A image (stores the compiler does not see)
__device__ __forceinline__ void wide_split_store_h(const f32x4 &u, const f32x4 &v, char *dst) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w0, w1;
  unsigned a, b;
  split_f16(u[0], u[1], a, b); w0[0] = a; w1[0] = b;
  split_f16(u[2], u[3], a, b); w0[1] = a; w1[1] = b;
  split_f16(v[0], v[1], a, b); w0[2] = a; w1[2] = b;
  split_f16(v[2], v[3], a, b); w0[3] = a; w1[3] = b;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)dst;
  static_assert(PLANE_BYTES == 8192, "plane offset below");
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(w0) : "memory");
  asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"(addr), "v"(w1) : "memory");
}

// The wide form (the bf16 kernel of round 2, removed, rebuilt on fp16 planes): 128 x 256 tile,
// eight multiplying waves, the same pipeline (B's images by LDS-DMA, A split by the multiplying waves between their matrix
// instructions, two LDS buffers, one raw barrier per k-tile, loads and DMAs counted by hand);
// per wave and k-tile 24 matrix instructions and 16 ds_read_b128 (48 and 24 on bf16 planes), 4 DMAs (6).
// The buffers are 48 KB, so the output staging area (64 KB) lies behind them instead of inside
// the buffer the last k-tile left free.
template <bool ACT>
__global__ __launch_bounds__(512) void gemm_wide_h_kernel(GemmArgs g) {
  extern __shared__ __align__(16) char smem[];        // 2 x (A image | 2 B images) | staging
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  constexpr int LA = ACT ? 6 : 2;                      // loads per thread and k-tile: A (+ coefficients)
  constexpr int NDMA = 4;                              // LDS-DMA instructions per wave and k-tile

  float sa, inv_a, sb, inv_b;
  pow2_scale(reduce_amax_slots(g.a_amax, lane), sa, inv_a);
  pow2_scale(reduce_amax_slots(g.b_amax, lane), sb, inv_b);

  const int tail_units = g.tail_splits > 1;
  const int nwg = tail_units ? (int)g.tail_first : (int)g.units;
  const int u_id = blockIdx.x;
  const bool tail = tail_units && u_id >= nwg;
  int tile, sp = 0;
  int64_t kchunk = g.K;
  if (tail) {
    const int v = u_id - nwg;
    tile = nwg + v / (int)g.tail_splits;
    sp = v % (int)g.tail_splits;
    kchunk = g.tail_kchunk;
  } else {
    const int q = nwg >> 3, rr = nwg & 7;
    const int xcd = u_id & 7, idx = u_id >> 3;
    tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  tile = __builtin_amdgcn_readfirstlane(tile);
  sp = __builtin_amdgcn_readfirstlane(sp);
  const int tn = tile % (int)g.tiles_n, tm = tile / (int)g.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * WBN;
  const int64_t kbeg = sp * kchunk;
  const int64_t kend = (kbeg + kchunk < g.K) ? (kbeg + kchunk) : g.K;
  const int64_t klen = kend - kbeg;
  const int ntf = __builtin_amdgcn_readfirstlane((int)(klen / BK));
  const int rem = __builtin_amdgcn_readfirstlane((int)(klen - (int64_t)ntf * BK));
  const int nt = ntf + (rem ? 1 : 0);
  const int nfull = (rem > 0 && rem <= 16) ? nt - 1 : nt;

  int64_t arow = t >> 2;
  if (m0 + arow >= g.M) arow = g.M - 1 - m0;
  const int kq = 8 * (t & 3);
  const unsigned a_off = (unsigned)((arow * g.lda + kq) * 4);
  const unsigned a_lds = plane_offset(t >> 2, 2 * (t & 3));
  const char *a_base = reinterpret_cast<const char *>(g.A + m0 * g.lda + kbeg);
  const char *cs_base = reinterpret_cast<const char *>(g.act_scale + kbeg);
  const char *ch_base = reinterpret_cast<const char *>(g.act_shift + kbeg);
  const char *b_img = g.b_planes + ((int64_t)tn * 2 * g.b_ktiles + kbeg / BK) * H_OPERAND_BYTES;
  const int64_t bs = g.b_ktiles * (int64_t)H_OPERAND_BYTES;

  f32x4 u, v, su, sv, hu, hv;
  auto load_a = [&](int kt) {
    const char *tp = a_base + (int64_t)kt * (BK * 4);
    unsigned ou = a_off, ov = a_off + 16, cu = 4u * kq, cv = 4u * kq + 16;
    if (kt >= ntf) {
      if (kq >= rem) { ou = a_off - 4u * kq; cu = 0; }
      if (kq + 4 >= rem) { ov = a_off - 4u * kq; cv = 0; }
    }
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(u) : "v"(ou), "s"(tp));
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(ov), "s"(tp));
    if constexpr (ACT) {
      const char *sp_ = cs_base + (int64_t)kt * (BK * 4), *hp_ = ch_base + (int64_t)kt * (BK * 4);
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(su) : "v"(cu), "s"(sp_));
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sv) : "v"(cv), "s"(sp_));
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(hu) : "v"(cu), "s"(hp_));
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(hv) : "v"(cv), "s"(hp_));
    }
  };
  auto split_a = [&](int kt) {
    if constexpr (ACT) asm volatile("" : "+v"(u), "+v"(v), "+v"(su), "+v"(sv), "+v"(hu), "+v"(hv));
    else asm volatile("" : "+v"(u), "+v"(v));
    if constexpr (ACT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = u[j] * su[j] + hu[j], b = v[j] * sv[j] + hv[j];
        u[j] = fmaxf(a, a * g.act_slope);
        v[j] = fmaxf(b, b * g.act_slope);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { u[j] *= sa; v[j] *= sa; }
    if (kt >= ntf) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kq + j >= rem) u[j] = 0.f;
        if (kq + 4 + j >= rem) v[j] = 0.f;
      }
    }
    wide_split_store_h(u, v, smem + (kt & 1) * H_BUF_BYTES + a_lds);
  };

  f32x16 hi[2][2], lo[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }

  wide_stage_b_h(b_img, bs, smem + H_OPERAND_BYTES, wave, lane);
  load_a(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  split_a(0);
  if (nt > 1) load_a(1);

  for (int it = 0; it < nfull; ++it) {
    const bool more = it + 1 < nt;
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (more)
      wide_stage_b_h(b_img + (int64_t)(it + 1) * H_OPERAND_BYTES, bs,
                     smem + ((it + 1) & 1) * H_BUF_BYTES + H_OPERAND_BYTES, wave, lane);
    const char *As = smem + (it & 1) * H_BUF_BYTES;
    const char *Bs = As + H_OPERAND_BYTES + (wn >> 1) * H_OPERAND_BYTES;
    const int bn0 = (wn & 1) * 64;
    f16x8 P[2], Q[2], R[2], S[2];
    read_set_h(As, wm * 64, lane, 0, P);
    read_set_h(Bs, bn0, lane, 0, Q);
    read_set_h(Bs, bn0 + 32, lane, 0, R);
    read_set_h(As, wm * 64 + 32, lane, 0, S);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma3(P, Q, hi[0][0], lo[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    mma3(P, R, hi[0][1], lo[0][1]);
    __builtin_amdgcn_sched_barrier(0);
    read_set_h(As, wm * 64, lane, 1, P);
    mma3(S, R, hi[1][1], lo[1][1]);
    __builtin_amdgcn_sched_barrier(0);
    read_set_h(Bs, bn0 + 32, lane, 1, R);
    mma3(S, Q, hi[1][0], lo[1][0]);
    __builtin_amdgcn_sched_barrier(0);
    read_set_h(Bs, bn0, lane, 1, Q);
    read_set_h(As, wm * 64 + 32, lane, 1, S);
    __builtin_amdgcn_s_setprio(0);
    if (more) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      split_a(it + 1);
      if (it + 2 < nt) load_a(it + 2);
    }
    mma3(P, R, hi[0][1], lo[0][1]);
    __builtin_amdgcn_sched_barrier(0);
    mma3(P, Q, hi[0][0], lo[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    mma3(S, Q, hi[1][0], lo[1][0]);
    __builtin_amdgcn_sched_barrier(0);
    mma3(S, R, hi[1][1], lo[1][1]);
  }
  if (nfull < nt) {                                    // the short last tile: first k-step only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char *As = smem + (nfull & 1) * H_BUF_BYTES;
    const char *Bs = As + H_OPERAND_BYTES + (wn >> 1) * H_OPERAND_BYTES;
    const int bn0 = (wn & 1) * 64;
    f16x8 P[2], Q[2], R[2], S[2];
    read_set_h(As, wm * 64, lane, 0, P);
    read_set_h(Bs, bn0, lane, 0, Q);
    read_set_h(Bs, bn0 + 32, lane, 0, R);
    read_set_h(As, wm * 64 + 32, lane, 0, S);
    mma3(P, Q, hi[0][0], lo[0][0]);
    mma3(P, R, hi[0][1], lo[0][1]);
    mma3(S, R, hi[1][1], lo[1][1]);
    mma3(S, Q, hi[1][0], lo[1][0]);
  }

  // C = (hi + 2^-11 lo) / (sa sb), the two inverse scales applied one after the other (their
  // product may leave the fp32 range where the result does not)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) hi[i][j][r] = ((hi[i][j][r] + lo[i][j][r] * H_LO) * inv_a) * inv_b;

  const int kh = lane >> 5, xl = lane & 31;
  if (tail) {
    float *slab = g.tail_slabs + ((int64_t)(tile - nwg) * g.tail_splits + sp) * (BM * WBN);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(wm * 64 + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2)) * WBN + wn * 64 + j * 32 + xl] = hi[i][j][r];
    return;
  }
  const int64_t col0 = n0 + wn * 64 + xl;
  float bv0 = (g.bias && col0 < g.N) ? g.bias[col0] : 0.f;
  float bv1 = (g.bias && col0 + 32 < g.N) ? g.bias[col0 + 32] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv0), "+v"(bv1)::"memory");
  if (g.c_vec && m0 + BM <= g.M) {
    // the wave transposes its block, 32 rows at a time, through its own 8 KB of the staging area
    // behind the plane buffers and stores 256-byte row segments, non-temporal
    float *ws = reinterpret_cast<float *>(smem + 2 * H_BUF_BYTES) + wave * (32 * 64);
    const int c = (lane & 15) * 4;
    const int64_t col = n0 + wn * 64 + c;
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && col < g.N) bq = *reinterpret_cast<const f32x4 *>(g.bias + col);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq)::"memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ws[(4 * kh + (r & 3) + 8 * (r >> 2)) * 64 + ((j * 32 + xl) ^ (kh << 5))] = hi[i][j][r];
      if (col < g.N) {
        float *orow = g.C + (m0 + wm * 64 + i * 32 + (lane >> 4)) * g.ldc + col;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 x = *reinterpret_cast<const f32x4 *>(&ws[(q * 4 + (lane >> 4)) * 64 + (c ^ ((q & 1) << 5))]);
          __builtin_nontemporal_store(x + bq, reinterpret_cast<f32x4 *>(orow + (int64_t)q * 4 * g.ldc));
        }
      }
    }
    return;
  }
  float *o = g.C + (m0 + wm * 64 + 4 * kh) * g.ldc + col0;
  const int64_t rbase = m0 + wm * 64 + 4 * kh;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bool colok = col0 + j * 32 < g.N;
    const float bv = j ? bv1 : bv0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
        if (colok && rbase + rr < g.M) o[rr * g.ldc + j * 32] = hi[i][j][r] + bv;
      }
  }
}

// four x of one k -> 8 bytes in each of the two fp16 planes of a k-major image
__device__ __forceinline__ void kmajor_split_store_h(const f32x4 &x, char *dst) {
  unsigned p01, p23, q01, q23;
  split_f16(x[0], x[1], p01, q01);
  split_f16(x[2], x[3], p23, q23);
  *reinterpret_cast<uint2 *>(dst) = make_uint2(p01, p23);
  *reinterpret_cast<uint2 *>(dst + PLANE_BYTES) = make_uint2(q01, q23);
}

__device__ __forceinline__ void read_set_tr_h(const char *img, int xbase, int lane, int step, f16x8 (&f)[2]) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int gq = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int ch = ((xbase + 16 * (gq & 1)) >> 3) + (p >> 1);
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    s16x4 h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = 16 * step + 8 * (gq >> 1) + 4 * hh + q;
      const char *a = img + pl * PLANE_BYTES + kmajor_offset(row, ch) + 8 * (p & 1);
      h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)a);
    }
    const s16x8 v = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
    f[pl] = __builtin_bit_cast(f16x8, v);
  }
}

// The wide form for k-major operands (the weight gradients), likewise rebuilt on fp16 planes:
// same units, same loads, same pipeline as its bf16 predecessor; both operands are scaled by their own power of two as they are split.
template <bool ACT>
__global__ __launch_bounds__(512) void gemm_kmajor_wide_h_kernel(GemmArgs g) {
  extern __shared__ __align__(16) char smem[];        // 2 x (A image | 2 B images), k-major
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  float sa, inv_a, sb, inv_b;
  pow2_scale(reduce_amax_slots(g.a_amax, lane), sa, inv_a);
  pow2_scale(reduce_amax_slots(g.b_amax, lane), sb, inv_b);
  const int nsp = (int)g.tail_splits;
  const int sp = __builtin_amdgcn_readfirstlane((int)blockIdx.x % nsp);
  const int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x / nsp);
  const int tn = tile % (int)g.tiles_n, tm = tile / (int)g.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * WBN;
  const int64_t kbeg = sp * g.tail_kchunk;
  const int64_t kend = (kbeg + g.tail_kchunk < g.K) ? (kbeg + g.tail_kchunk) : g.K;
  const int ntf = __builtin_amdgcn_readfirstlane((int)((kend - kbeg) / BK));
  const int rem = __builtin_amdgcn_readfirstlane((int)(kend - kbeg - (int64_t)ntf * BK));

  const int ax = 4 * (t & 31), ak = t >> 5;
  const int bx = 4 * (t & 63), bk = t >> 6;
  const int axc = (m0 + ax < g.M) ? ax : 0, bxc = (n0 + bx < g.N) ? bx : 0;
  const char *a_base = reinterpret_cast<const char *>(g.A + kbeg * g.lda + m0);
  const char *b_base = reinterpret_cast<const char *>(g.B + kbeg * g.ldb + n0);
  auto krow = [&](int j) { return j < 2 ? ak + 16 * j : bk + 8 * (j - 2); };
  unsigned off[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
    off[j] = j < 2 ? (unsigned)((krow(j) * g.lda + axc) * 4) : (unsigned)((krow(j) * g.ldb + bxc) * 4);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if constexpr (ACT) {
    // (the operand's power-of-two scale rides in the coefficients: lrelu is positively
    // homogeneous and a power of two commutes with every rounding, so lrelu(v sc + sh) sb =
    // lrelu(v (sc sb) + sh sb) bit for bit - one multiplication less per element and k-tile)
    sc = *reinterpret_cast<const f32x4 *>(g.act_scale + n0 + bxc) * sb;
    sh = *reinterpret_cast<const f32x4 *>(g.act_shift + n0 + bxc) * sb;
  }
  f32x4 r[6];
  auto tile_a = [&](int kt) { return a_base + (int64_t)kt * BK * g.lda * 4; };
  auto tile_b = [&](int kt) { return b_base + (int64_t)kt * BK * g.ldb * 4; };
  auto split_piece = [&](int j, char *buf) {
    if (j < 2) {
      const f32x4 v = r[j] * sa;
      kmajor_split_store_h(v, buf + kmajor_offset(krow(j), ax >> 3) + 2 * (ax & 7));
    } else {
      f32x4 v = r[j];
      if constexpr (ACT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = v[e] * sc[e] + sh[e];
          v[e] = fmaxf(a, a * g.act_slope);
        }
      } else {
        v *= sb;
      }
      kmajor_split_store_h(v, buf + H_OPERAND_BYTES + (bx >> 7) * H_OPERAND_BYTES +
                                  kmajor_offset(krow(j), (bx & 127) >> 3) + 2 * (bx & 7));
    }
  };

  f32x16 hi[2][2], lo[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) { hi[i][j][q] = 0.f; lo[i][j][q] = 0.f; }

  // one k-tile out of buffer `As`; piece(j) runs after the j-th block of matrix instructions
  auto multiply = [&](const char *As, auto &&piece) {
    const char *Bs = As + H_OPERAND_BYTES + (wn >> 1) * H_OPERAND_BYTES;
    const int bn0 = (wn & 1) * 64;
    f16x8 P[2], Q[2], R[2], S[2];
    read_set_tr_h(As, wm * 64, lane, 0, P);
    read_set_tr_h(Bs, bn0, lane, 0, Q);
    read_set_tr_h(Bs, bn0 + 32, lane, 0, R);
    read_set_tr_h(As, wm * 64 + 32, lane, 0, S);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma3(P, Q, hi[0][0], lo[0][0]);
    piece(0);
    __builtin_amdgcn_sched_barrier(0);
    mma3(P, R, hi[0][1], lo[0][1]);
    piece(1);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr_h(As, wm * 64, lane, 1, P);
    mma3(S, R, hi[1][1], lo[1][1]);
    piece(2);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr_h(Bs, bn0 + 32, lane, 1, R);
    mma3(S, Q, hi[1][0], lo[1][0]);
    piece(3);
    __builtin_amdgcn_sched_barrier(0);
    read_set_tr_h(Bs, bn0, lane, 1, Q);
    read_set_tr_h(As, wm * 64 + 32, lane, 1, S);
    __builtin_amdgcn_s_setprio(0);
    mma3(P, R, hi[0][1], lo[0][1]);
    piece(4);
    __builtin_amdgcn_sched_barrier(0);
    mma3(P, Q, hi[0][0], lo[0][0]);
    piece(5);
    __builtin_amdgcn_sched_barrier(0);
    mma3(S, Q, hi[1][0], lo[1][0]);
    __builtin_amdgcn_sched_barrier(0);
    mma3(S, R, hi[1][1], lo[1][1]);
  };
  auto nothing = [](int) {};

  if (ntf > 0) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[j]) : "v"(off[j]), "s"(j < 2 ? tile_a(0) : tile_b(0)));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5])::"memory");
#pragma unroll
    for (int j = 0; j < 6; ++j) split_piece(j, smem);
    const int t1 = ntf > 1 ? 1 : 0;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[j]) : "v"(off[j]), "s"(j < 2 ? tile_a(t1) : tile_b(t1)));
    for (int it = 0; it + 1 < ntf; ++it) {
      __syncthreads();
      char *other = smem + ((it + 1) & 1) * H_BUF_BYTES;
      const int kt = it + 2 < ntf ? it + 2 : ntf - 1;
      const char *ta = tile_a(kt), *tb = tile_b(kt);
      multiply(smem + (it & 1) * H_BUF_BYTES, [&](int j) {
        asm volatile("s_waitcnt vmcnt(5)" : "+v"(r[j]));
        split_piece(j, other);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[j]) : "v"(off[j]), "s"(j < 2 ? ta : tb));
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (rem > 0) {
    const char *ta = tile_a(ntf), *tb = tile_b(ntf);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int kr = krow(j) < rem ? krow(j) : rem - 1;
      const unsigned o = j < 2 ? (unsigned)((kr * g.lda + axc) * 4) : (unsigned)((kr * g.ldb + bxc) * 4);
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[j]) : "v"(o), "s"(j < 2 ? ta : tb));
    }
  }
  if (ntf > 0) {
    __syncthreads();
    multiply(smem + ((ntf - 1) & 1) * H_BUF_BYTES, nothing);
  }
  if (rem > 0) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5])::"memory");
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (krow(j) >= rem) r[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    char *buf = smem + (ntf & 1) * H_BUF_BYTES;
#pragma unroll
    for (int j = 0; j < 6; ++j) split_piece(j, buf);
    __syncthreads();
    multiply(buf, nothing);
  }

  const int kh = lane >> 5, xl = lane & 31;
  float *slab = g.tail_slabs + ((int64_t)tile * nsp + sp) * (BM * WBN);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        slab[(wm * 64 + i * 32 + 4 * kh + (q & 3) + 8 * (q >> 2)) * WBN + wn * 64 + j * 32 + xl] =
            ((hi[i][j][q] + lo[i][j][q] * H_LO) * inv_a) * inv_b;
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
    const int64_t row = (int64_t)tm * BM + rl;
    if (row >= g.M) continue;
    const float rs = g.row_scale ? g.row_scale[row] : 1.0f;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = (int64_t)tn * BN + c4 + j;
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

// workgroups resident at once: one per CU (96 KB of LDS each)
int64_t resident_workgroups() {
  static int64_t slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    slots = (int64_t)cus;
  }
  return slots;
}

struct Plan {
  int64_t splits, k_chunk;                          // whole-problem split-K
  int64_t tail_first, tail_splits, tail_kchunk;     // tail split (mutually exclusive with it)
  int64_t slab_bytes;                               // slabs of either split, at the workspace's start
  int64_t planes_bytes;                             // pre-split B behind them (0: split in the kernel)
  int64_t workspace_bytes;
  // the wide form (128 x 256 tiles, gemm_wide_h_kernel): taken when the shape allows it and the
  // operands qualify (run_gemm); its tail split has slabs of its own size
  bool wide;
  int64_t w_tiles, w_tail_first, w_tail_splits, w_tail_kchunk, w_slab_bytes;
  // the wide form for k-major operands (gemm_kmajor_wide_h_kernel): every tile split over K into
  // one round of the chip; 0 splits = not taken
  int64_t km_splits, km_kchunk, km_slab_bytes;
  int64_t amax_bytes;                               // 2 x 256 partial maxima behind the planes
};

int64_t choose_splits(int64_t m, int64_t n, int64_t k);

Plan make_plan(int64_t m, int64_t n, int64_t k) {
  Plan p;
  const int64_t tiles = tv_cdiv(m, BM) * tv_cdiv(n, BN);
  p.splits = choose_splits(m, n, k);
  p.k_chunk = tv_cdiv(tv_cdiv(k, p.splits), BK) * BK;
  p.splits = tv_cdiv(k, p.k_chunk);
  p.tail_first = tiles; p.tail_splits = 1; p.tail_kchunk = p.k_chunk;
  p.slab_bytes = p.splits > 1 ? p.splits * m * n * (int64_t)sizeof(float) : 0;
  const int64_t slots = resident_workgroups();
  if (p.splits == 1 && tiles > slots && tiles < (1ll << 30)) {
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
          p.slab_bytes = tail * ts * (int64_t)(BM * BN * sizeof(float));
        }
      }
    }
  }
  // B is split ahead of the product when it is small and every piece of it is needed by many
  // row tiles (a layer's weights under >= 2048 rows): 6 bytes per element of scratch
  p.planes_bytes = 0;
  // (an even number of 128-column images: the wide form takes them two at a time)
  if (m >= 2048 && n * k <= (1ll << 25))
    p.planes_bytes = ((tv_cdiv(n, BN) + 1) & ~1ll) * tv_cdiv(k, BK) * (int64_t)OPERAND_BYTES;
  // the wide form: enough 128 x 256 tiles for one round of the chip, B pre-split, no split-K
  p.w_tiles = tv_cdiv(m, BM) * tv_cdiv(n, WBN);
  p.wide = p.planes_bytes > 0 && p.splits == 1 && p.w_tiles >= slots && p.w_tiles < (1ll << 30);
  p.w_tail_first = p.w_tiles; p.w_tail_splits = 1; p.w_tail_kchunk = tv_cdiv(k, BK) * BK; p.w_slab_bytes = 0;
  if (p.wide) {
    const int64_t tail = p.w_tiles % slots;
    if (tail > 0) {
      int64_t ts = slots / tail;
      const int64_t kmax = k / (4 * BK);
      if (ts > kmax) ts = kmax;
      if (ts > 8) ts = 8;
      if (ts >= 2) {
        const int64_t kc = tv_cdiv(tv_cdiv(k, ts), BK) * BK;
        ts = tv_cdiv(k, kc);
        if (ts >= 2) {
          p.w_tail_first = p.w_tiles - tail; p.w_tail_splits = ts; p.w_tail_kchunk = kc;
          p.w_slab_bytes = tail * ts * (int64_t)(BM * WBN * sizeof(float));
        }
      }
    }
  }
  if (p.w_slab_bytes > p.slab_bytes) p.slab_bytes = p.w_slab_bytes;   // either form may be taken
  p.km_splits = 0; p.km_kchunk = 0; p.km_slab_bytes = 0;
  if (p.planes_bytes == 0 && n > BN && 2 * p.w_tiles <= slots && p.w_tiles > 0) {
    int64_t ks = slots / p.w_tiles;                   // floor: never a second, partial round
    const int64_t kmax = k / (2 * BK);
    if (ks > kmax) ks = kmax;
    if (ks > 64) ks = 64;
    if (ks >= 2) {
      p.km_kchunk = tv_cdiv(tv_cdiv(k, ks), BK) * BK;
      p.km_splits = tv_cdiv(k, p.km_kchunk);
      p.km_slab_bytes = p.w_tiles * p.km_splits * (int64_t)(BM * WBN * sizeof(float));
      if (p.km_splits < 2) { p.km_splits = 0; p.km_slab_bytes = 0; }
    }
  }
  if (p.km_slab_bytes > p.slab_bytes) p.slab_bytes = p.km_slab_bytes;
  p.slab_bytes = (p.slab_bytes + 255) & ~255ll;
  p.amax_bytes = (p.wide || p.km_splits > 1) ? 2 * AMAX_SLOTS * (int64_t)sizeof(float) : 0;
  p.workspace_bytes = p.slab_bytes + p.planes_bytes + p.amax_bytes;
  return p;
}

int64_t choose_splits(int64_t m, int64_t n, int64_t k) {
  const int64_t tiles = tv_cdiv(m, BM) * tv_cdiv(n, BN);
  if (tiles >= 256 || k < 4 * BK) return 1;
  // one round of the persistent grid: more, shorter units only add slab traffic and epilogues
  // (512 units -> 256: +2.7 % on the 1000 x 1000 x 51200 weight gradients)
  // (floor: 208 tiles split in two are 416 units, a full round and a partial one of half the
  // length each - the time of the one unsplit round plus the slabs and their reduction)
  int64_t s = resident_workgroups() / tiles;
  // a workgroup alone on its CU pays the full memory latency per k-tile, so few-tile problems
  // are split deep: at least 2 k-tiles per split
  constexpr int64_t min_tiles = 2;
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
thread_local int g_last_form = 0;

// Measurement aid (tv_gemm_profile_begin / _end): while it is on, the MAIN kernel of every product
// is bracketed by a pair of HIP events on the stream it is launched on - the kernel alone,
// without the operand scans, the split of B and the reductions around it.
struct GemmProfileRecord { hipEvent_t start, stop; int form; int64_t m, n, k; };
struct GemmProfile {
  std::mutex lock;                  // forward and backward passes come from different threads
  bool on = false;
  std::vector<GemmProfileRecord> records;
} g_profile;
struct ProfileScope {
  bool active = false;
  GemmProfileRecord rec;
  hipStream_t stream;
  ProfileScope(hipStream_t s, int form, int64_t m, int64_t n, int64_t k) : stream(s) {
    std::lock_guard<std::mutex> guard(g_profile.lock);
    if (!g_profile.on) return;
    rec.form = form; rec.m = m; rec.n = n; rec.k = k;
    if (hipEventCreate(&rec.start) != hipSuccess) return;
    if (hipEventCreate(&rec.stop) != hipSuccess) { (void)hipEventDestroy(rec.start); return; }
    (void)hipEventRecord(rec.start, stream);
    active = true;
  }
  void done() {
    if (!active) return;
    (void)hipEventRecord(rec.stop, stream);
    std::lock_guard<std::mutex> guard(g_profile.lock);
    g_profile.records.push_back(rec);
    active = false;
  }
};

int run_gemm(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float *a,
             int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc, const float *bias,
             const float *row_scale, int accumulate, int act_operand, const float *act_scale,
             const float *act_shift, float act_slope, void *workspace, int64_t workspace_bytes,
             void *stream, float *a_max = nullptr, int a_max_state = 0, float *b_max = nullptr,
             int b_max_state = 0) {
  TV_REQUIRE(m >= 0 && n >= 0 && k >= 0, "tv_gemm_f32: negative size");
  TV_REQUIRE(a_max_state >= 0 && a_max_state <= 2 && b_max_state >= 0 && b_max_state <= 4 &&
                 (a_max || a_max_state == 0) && (b_max || b_max_state == 0 || b_max_state == 3),
             "tv_gemm_f32_ex: operand-maximum state must be 0 (none), 1 (compute into the buffer) "
             "or 2 (the buffer is valid), with a buffer for 1 and 2; for B also 3 / 4 (b is a "
             "plane image from tv_gemm_prepare_b_batch: bf16 / fp16 form)");
  // B handed over as a ready-made plane image (tv_gemm_prepare_b_batch)
  const int b_prepared = b_max_state >= 3 ? (b_max_state == 4 ? 1 : 0) : -1;
  if (b_max_state == 3) b_max_state = 0;
  if (b_max_state == 4) b_max_state = 2;
  if (m == 0 || n == 0) return TV_OK;
  TV_REQUIRE(a && b && c, "tv_gemm_f32: null operand");
  TV_REQUIRE(lda >= (trans_a ? m : k), "tv_gemm_f32: lda %lld too small", (long long)lda);
  TV_REQUIRE(ldb >= (trans_b ? k : n), "tv_gemm_f32: ldb %lld too small", (long long)ldb);
  TV_REQUIRE(ldc >= n, "tv_gemm_f32: ldc %lld too small", (long long)ldc);
  TV_REQUIRE(k > 0, "tv_gemm_f32: k must be positive");
  TV_REQUIRE(lda < (1ll << 22) && ldb < (1ll << 22), "tv_gemm_f32: leading dimension too large");

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
    TV_REQUIRE(act_operand != 1 || k >= 4, "tv_gemm_f32_act: the A prologue needs k >= 4");
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
    } else if (g.tail_splits > 1) {
      g.tail_slabs = (float *)workspace;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  // the small operand split once, ahead of the product (not with a prologue on B: that is the
  // weight-gradient form, whose B is the activation matrix)
  const bool presplit = plan.planes_bytes > 0 && act_operand != 2;
  g.b_planes = nullptr; g.b_ktiles = tv_cdiv(k, BK);
  g.a_amax = g.b_amax = nullptr;
  float *amax_a = nullptr, *amax_b = nullptr;
  if (plan.amax_bytes > 0) {
    amax_a = (float *)((char *)workspace + plan.slab_bytes + plan.planes_bytes);
    amax_b = amax_a + AMAX_SLOTS;
  }
  // caller-kept maxima (tv_gemm_f32_ex): computed into the caller's buffer, or taken from it
  if (a_max_state) amax_a = a_max;
  if (b_max_state) amax_b = b_max;
  // max |operand| of a row-major [rows][cols] operand (behind its prologue), 256 partial maxima
  auto launch_amax = [&](const float *x, int64_t ld, int64_t rows, int64_t cols, bool act, float *slots) {
    const int vec = (ld % 4 == 0) && (((uintptr_t)x & 15) == 0) &&
                    (!act || ((((uintptr_t)act_scale | (uintptr_t)act_shift) & 15) == 0));
    if (act)
      hipLaunchKernelGGL(gemm_amax_kernel<true>, dim3(AMAX_SLOTS), dim3(1024), 0, s, x, ld, rows, cols, vec,
                         act_scale, act_shift, act_slope, slots);
    else
      hipLaunchKernelGGL(gemm_amax_kernel<false>, dim3(AMAX_SLOTS), dim3(1024), 0, s, x, ld, rows, cols, vec,
                         (const float *)nullptr, (const float *)nullptr, 0.f, slots);
  };
  // the wide form: a row-major A that can be read 16 bytes at a time against the pre-split B,
  // plain epilogue (bias at most)
  // (a K that is not a multiple of 4 is taken too when the rows of A are padded to one: the float4
  // that straddles K is readable, its elements beyond K are zeroed once landed and B's planes are
  // zero there anyway; not with a prologue, whose coefficient vectors end at K)
  const bool a_rows16 = g.a_vec || (!trans_a && act_operand == 0 && lda % 4 == 0 &&
                                    (((uintptr_t)a & 15) == 0) && lda >= ((k + 3) & ~3ll));
  // (not for the 40 / 50-wide first layers: with n or k below 64 the product is bound by the
  // bytes of its tall operand or of its output, and the producer / consumer form, whose stores
  // overlap the next unit, streams them faster - 51200 x 40 x 1004: 123 -> 90 us, 51200 x 1004 x
  // 40: 79 -> 70 us, tools/narrow_gemm_lab.py)
  const bool take_wide = plan.wide && presplit && !trans_a && a_rows16 && !row_scale && !accumulate &&
                         act_operand != 2 && k >= 64 && n >= 64;
  if (b_prepared >= 0) {
    TV_REQUIRE(presplit && b_prepared == (take_wide ? 1 : 0),
               "tv_gemm_f32_ex: the prepared B image is of form %d but this product (m %lld n %lld k %lld) "
               "takes %s", b_prepared, (long long)m, (long long)n, (long long)k,
               !presplit ? "no pre-split B" : (take_wide ? "the fp16 form" : "the bf16 form"));
    TV_REQUIRE(((uintptr_t)b & 15) == 0, "tv_gemm_f32_ex: the prepared B image must be 16-byte aligned");
    g.b_planes = reinterpret_cast<const char *>(b);
  } else if (presplit) {
    TV_REQUIRE(((uintptr_t)workspace & 15) == 0, "tv_gemm_f32: workspace must be 16-byte aligned");
    char *planes = (char *)workspace + plan.slab_bytes;
    g.b_planes = planes;
    const dim3 sgrid((unsigned)(((g.tiles_n + 1) & ~1ll) * g.b_ktiles));
    if (take_wide) {
      // B as stored: [n][k] (trans_b) or [k][n]
      if (b_max_state != 2) launch_amax(b, ldb, trans_b ? n : k, trans_b ? k : n, false, amax_b);
      TV_LAUNCH_CHECK("tv_gemm_f32(max of B)");
      hipLaunchKernelGGL(gemm_split_b_h_kernel, sgrid, dim3(256), 0, s, b, ldb, trans_b ? 0 : 1, n, k,
                         g.b_ktiles, (const float *)amax_b, planes);
    } else {
      hipLaunchKernelGGL(gemm_split_b_kernel, sgrid, dim3(256), 0, s, b, ldb, trans_b ? 0 : 1, n, k,
                         g.b_ktiles, planes);
    }
    TV_LAUNCH_CHECK("tv_gemm_f32(split of B)");
  }
  if (take_wide) {
    GemmArgs w = g;
    w.tiles_n = tv_cdiv(n, WBN);
    w.tail_first = plan.w_tail_first; w.tail_splits = plan.w_tail_splits; w.tail_kchunk = plan.w_tail_kchunk;
    w.tail_slabs = plan.w_tail_splits > 1 ? (float *)workspace : nullptr;
    const int64_t units = plan.w_tail_splits > 1
                              ? plan.w_tail_first + (plan.w_tiles - plan.w_tail_first) * plan.w_tail_splits
                              : plan.w_tiles;
    TV_REQUIRE(units < (1ll << 31), "tv_gemm_f32: grid too large");
    w.units = units;
    w.a_amax = amax_a; w.b_amax = amax_b;
    if (a_max_state != 2) launch_amax(a, lda, m, k, act_operand == 1, amax_a);
    TV_LAUNCH_CHECK("tv_gemm_f32(max of A)");
    // two plane buffers + the eight waves' output staging behind them
    const size_t wlds = 2 * (size_t)H_BUF_BYTES + 8 * 32 * 64 * sizeof(float);
    int device = 0;
    TV_HIP(hipGetDevice(&device));
    static bool raised[2][64] = {};
    const int wa = act_operand == 1;
    if (!raised[wa][device & 63]) {
      if (wa) TV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_wide_h_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
      else TV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_wide_h_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
      raised[wa][device & 63] = true;
    }
    ProfileScope prof(s, 1, m, n, k);
    if (wa) hipLaunchKernelGGL(gemm_wide_h_kernel<true>, dim3((unsigned)units), dim3(512), wlds, s, w);
    else hipLaunchKernelGGL(gemm_wide_h_kernel<false>, dim3((unsigned)units), dim3(512), wlds, s, w);
    prof.done();
    TV_LAUNCH_CHECK("tv_gemm_f32(wide)");
    if (w.tail_splits > 1) {
      hipLaunchKernelGGL(gemm_wide_tail_reduce_kernel, dim3((unsigned)((plan.w_tiles - w.tail_first) * 8)),
                         dim3(256), 0, s, w);
      TV_LAUNCH_CHECK("tv_gemm_f32(wide tail reduce)");
    }
    g_last_form = 1;
    return TV_OK;
  }
  // the wide form for k-major operands (weight gradients): both operands readable 16 bytes at a
  // time along x, plain epilogue (bias at most), B prologue allowed
  // (an N that is not a multiple of 4 is taken when the rows of B are padded to one and there is
  // no prologue on it: the float4 that straddles N is readable and only reaches accumulators that
  // are not stored - the class-count gradient of the logits layer, 1001 columns in rows of 1004)
  const bool b_cols16 = g.b_vec || (!trans_b && act_operand == 0 && ldb % 4 == 0 &&
                                    (((uintptr_t)b & 15) == 0) && ldb >= ((n + 3) & ~3ll));
  if (plan.km_splits > 1 && trans_a && !trans_b && g.a_vec && b_cols16 && !presplit && !row_scale &&
      !accumulate && act_operand != 1) {
    TV_REQUIRE(((uintptr_t)workspace & 15) == 0, "tv_gemm_f32: workspace must be 16-byte aligned");
    GemmArgs w = g;
    w.tiles_n = tv_cdiv(n, WBN);
    w.splits = 1; w.slabs = nullptr;
    w.tail_first = 0; w.tail_splits = plan.km_splits; w.tail_kchunk = plan.km_kchunk;
    w.tail_slabs = (float *)workspace;
    const int64_t units = plan.w_tiles * plan.km_splits;
    w.units = units;
    w.a_amax = amax_a; w.b_amax = amax_b;
    if (a_max_state != 2) launch_amax(a, lda, k, m, false, amax_a);                    // A stored [k][m]
    if (b_max_state != 2) launch_amax(b, ldb, k, n, act_operand == 2, amax_b);         // B stored [k][n]
    TV_LAUNCH_CHECK("tv_gemm_f32(max of the operands)");
    const size_t wlds = 2 * (size_t)H_BUF_BYTES;
    int device = 0;
    TV_HIP(hipGetDevice(&device));
    static bool raised[2][64] = {};
    const int wa = act_operand == 2;
    if (!raised[wa][device & 63]) {
      if (wa) TV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kmajor_wide_h_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
      else TV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kmajor_wide_h_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
      raised[wa][device & 63] = true;
    }
    ProfileScope prof(s, 2, m, n, k);
    if (wa) hipLaunchKernelGGL(gemm_kmajor_wide_h_kernel<true>, dim3((unsigned)units), dim3(512), wlds, s, w);
    else hipLaunchKernelGGL(gemm_kmajor_wide_h_kernel<false>, dim3((unsigned)units), dim3(512), wlds, s, w);
    prof.done();
    TV_LAUNCH_CHECK("tv_gemm_f32(k-major wide)");
    hipLaunchKernelGGL(gemm_wide_tail_reduce_kernel, dim3((unsigned)(plan.w_tiles * 8)), dim3(256), 0, s, w);
    TV_LAUNCH_CHECK("tv_gemm_f32(k-major wide reduce)");
    g_last_form = 2;
    return TV_OK;
  }
  g_last_form = 0;
  const int64_t tiles = g.tiles_m * g.tiles_n;
  const int64_t nwg = g.tail_splits > 1 ? g.tail_first + (tiles - g.tail_first) * g.tail_splits
                                        : tiles * g.splits;
  TV_REQUIRE(nwg < (1ll << 31), "tv_gemm_f32: grid too large");
  g.units = nwg;
  // persistent grid: one workgroup per CU (a multiple of 8, so that a workgroup's units stay on
  // the tile range of its XCD), or one per unit when there are fewer
  int64_t resident = resident_workgroups() & ~7ll;
  if (resident < 8) resident = 8;
  const dim3 grid((unsigned)(nwg < resident ? nwg : resident)), block(512);
  // two buffers x (A planes | B planes) + the consumers' epilogue staging
  const size_t lds = 4 * (size_t)OPERAND_BYTES + STAGE_BYTES;
  int device = 0;
  TV_HIP(hipGetDevice(&device));
  // load mode of each operand: 0 = [x][k] 16-byte loads, 1 = [x][k] 4-byte loads, 2 = [k][x]
  const int am = trans_a ? 2 : (g.a_vec ? 0 : 1);
  const int bm = presplit ? 3 : trans_b ? (g.b_vec ? 0 : 1) : 2;
#define TV_GEMM(AM, BM_, ACT)                                                                    \
  do {                                                                                           \
    static bool raised[64] = {};  /* 160 KB of dynamic LDS: once per kernel and device */       \
    if (!raised[device & 63]) {                                                                  \
      TV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kernel<AM, BM_, ACT>),     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
      raised[device & 63] = true;                                                                \
    }                                                                                            \
    hipLaunchKernelGGL((gemm_kernel<AM, BM_, ACT>), grid, block, lds, s, g);                     \
  } while (0)
#define TV_GEMM_B(AM, ACT)                                                                       \
  do {                                                                                           \
    if (bm == 0) TV_GEMM(AM, 0, ACT); else if (bm == 1) TV_GEMM(AM, 1, ACT);                     \
    else if (bm == 2) TV_GEMM(AM, 2, ACT); else TV_GEMM(AM, 3, ACT);                             \
  } while (0)
  ProfileScope prof(s, 0, m, n, k);
  if (act_operand == 1) {            // A row-major
    if (am == 0) TV_GEMM_B(0, 1); else TV_GEMM_B(1, 1);
  } else if (act_operand == 2) {     // B k-major
    if (am == 0) TV_GEMM(0, 2, 2); else if (am == 1) TV_GEMM(1, 2, 2); else TV_GEMM(2, 2, 2);
  } else {
    if (am == 0) TV_GEMM_B(0, 0); else if (am == 1) TV_GEMM_B(1, 0); else TV_GEMM_B(2, 0);
  }
#undef TV_GEMM_B
#undef TV_GEMM
  prof.done();
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

extern "C" int tv_gemm_f32_last_form(void) { return g_last_form; }

// ------------------------------------------------------------------------------------
// B prepared ahead of the products (tv_gemm_prepare_b_batch): the weights of a stack of layers
// change once per step, so their plane images - one per orientation: the forward product reads
// W as [n][k], the input gradient as [k][n] - are all written by ONE launch instead of one
// launch in front of every product.
// ------------------------------------------------------------------------------------
extern "C" int tv_gemm_b_form(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return -1;
  const Plan plan = make_plan(m, n, k);
  if (plan.planes_bytes == 0) return -1;             // B is split inside the kernel
  return (plan.wide && k >= 64 && n >= 64) ? 1 : 0;  // as run_gemm decides for a plain product
}

extern "C" int64_t tv_gemm_prepared_b_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  return make_plan(m, n, k).planes_bytes;
}

extern "C" int tv_gemm_prepare_b_batch(int64_t count, const int64_t *m, const int64_t *n,
                                       const int64_t *k, const int *trans_b,
                                       const float *const *b, const int64_t *ldb,
                                       const float *const *b_max, void *const *planes,
                                       void *stream) {
  TV_REQUIRE(count >= 0 && count <= SPLIT_BATCH, "tv_gemm_prepare_b_batch: at most %d operands per call",
             SPLIT_BATCH);
  if (count == 0) return TV_OK;
  TV_REQUIRE(m && n && k && trans_b && b && ldb && b_max && planes, "tv_gemm_prepare_b_batch: null array");
  SplitBatch batch = {};
  int64_t most = 0;
  for (int64_t i = 0; i < count; ++i) {
    const int form = tv_gemm_b_form(m[i], n[i], k[i]);
    TV_REQUIRE(form >= 0, "tv_gemm_prepare_b_batch: operand %lld (m %lld n %lld k %lld) is not pre-split",
               (long long)i, (long long)m[i], (long long)n[i], (long long)k[i]);
    TV_REQUIRE(b[i] && planes[i] && (((uintptr_t)planes[i]) & 15) == 0,
               "tv_gemm_prepare_b_batch: null operand or unaligned image");
    TV_REQUIRE(ldb[i] >= (trans_b[i] ? k[i] : n[i]), "tv_gemm_prepare_b_batch: ldb too small");
    TV_REQUIRE(form == 0 || b_max[i], "tv_gemm_prepare_b_batch: the fp16 form needs the operand's maxima");
    SplitItem &it = batch.item[i];
    it.b = b[i]; it.slots = b_max[i]; it.planes = (char *)planes[i];
    it.ldb = ldb[i]; it.n = n[i]; it.k = k[i]; it.ktiles = tv_cdiv(k[i], BK);
    it.blocks = ((tv_cdiv(n[i], BN) + 1) & ~1ll) * it.ktiles;
    it.kx = trans_b[i] ? 0 : 1; it.form = form;
    if (it.blocks > most) most = it.blocks;
  }
  TV_REQUIRE(most < (1ll << 31), "tv_gemm_prepare_b_batch: operand too large");
  hipLaunchKernelGGL(gemm_split_b_batch_kernel, dim3((unsigned)most, (unsigned)count), dim3(256), 0,
                     (hipStream_t)stream, batch);
  TV_LAUNCH_CHECK("tv_gemm_prepare_b_batch");
  return TV_OK;
}

extern "C" int tv_gemm_profile_begin(void) {
  std::lock_guard<std::mutex> guard(g_profile.lock);
  for (auto &r : g_profile.records) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  g_profile.records.clear();
  g_profile.on = true;
  return TV_OK;
}

extern "C" int64_t tv_gemm_profile_end(double *records, int64_t capacity) {
  std::lock_guard<std::mutex> guard(g_profile.lock);
  g_profile.on = false;
  int64_t count = 0;
  for (auto &r : g_profile.records) {
    float ms = 0.f;
    const bool ok = hipEventSynchronize(r.stop) == hipSuccess &&
                    hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess;
    if (ok && records && count < capacity) {
      double *o = records + 5 * count;
      o[0] = (double)ms; o[1] = (double)r.form; o[2] = (double)r.m; o[3] = (double)r.n; o[4] = (double)r.k;
    }
    if (ok) ++count;
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  g_profile.records.clear();
  return count;
}

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

extern "C" int tv_gemm_f32_ex(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k,
                              const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                              int64_t ldc, const float *bias, const float *row_scale,
                              int accumulate, int act_operand, const float *act_scale,
                              const float *act_shift, float act_slope, float *a_max,
                              int a_max_state, float *b_max, int b_max_state, void *workspace,
                              int64_t workspace_bytes, void *stream) {
  return run_gemm(trans_a, trans_b, m, n, k, a, lda, b, ldb, c, ldc, bias, row_scale, accumulate,
                  act_operand, act_scale, act_shift, act_slope, workspace, workspace_bytes, stream,
                  a_max, a_max_state, b_max, b_max_state);
}
