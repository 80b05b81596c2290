// GraphConv aggregation, graph readout, embedding gather / segmented scatter and the
// latency-decoder input rows.  All HBM-bound gathers over the batching kernel's CSC /
// segment offsets; every sum runs in a fixed order (no float atomics).
#include "tv_common.h"

namespace {

__device__ __forceinline__ float deg_norm(const int64_t *indptr, int64_t v) {
  const float d = (float)(indptr[v + 1] - indptr[v]);
  return 1.0f / sqrtf(fmaxf(d, 1.0f));
}

// wave per destination row, float4 chunks (cols % 4 == 0)
__global__ __launch_bounds__(256) void graph_sum_vec_kernel(
    int64_t rows_per_rep, int64_t rows, int64_t c4, const int64_t *__restrict__ indptr,
    const int64_t *__restrict__ indices, const float *__restrict__ x, int scale_src,
    int scale_dst, float *__restrict__ out, float *out_max,
    const float *__restrict__ act_scale, const float *__restrict__ act_shift, float act_slope) {
  const int lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * nw + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * nw;
  float omax = 0.f;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t rep = r / rows_per_rep, v = r - rep * rows_per_rep, base = rep * rows_per_rep;
    const int64_t p0 = indptr[v], p1 = indptr[v + 1];
    const float nd = scale_dst ? deg_norm(indptr, v) : 1.0f;
    for (int64_t q0 = 0; q0 < c4; q0 += 64 * 4) {
      float4 acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      // optional prologue: the rows are the pre-normalisation output of a GAT layer and go through
      // lrelu(v * scale[col] + shift[col]) as they are gathered (the BatchNorm + LeakyReLU that
      // the reference applies before this layer; its result is never written)
      float4 psc[4], psh[4];
      if (act_scale) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t q = q0 + lane + 64 * c;
          psc[c] = (q < c4) ? reinterpret_cast<const float4 *>(act_scale)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          psh[c] = (q < c4) ? reinterpret_cast<const float4 *>(act_shift)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      auto act = [&](float4 v, int c) -> float4 {
        if (!act_scale) return v;
        return make_float4(tv_lrelu(fmaf(v.x, psc[c].x, psh[c].x), act_slope),
                           tv_lrelu(fmaf(v.y, psc[c].y, psh[c].y), act_slope),
                           tv_lrelu(fmaf(v.z, psc[c].z, psh[c].z), act_slope),
                           tv_lrelu(fmaf(v.w, psc[c].w, psh[c].w), act_slope));
      };
      // two neighbour rows in flight per trip (the rows are still added in edge order)
      int64_t p = p0;
      for (; p + 2 <= p1; p += 2) {
        const int64_t u0 = indices[p], u1 = indices[p + 1];
        const float ns0 = scale_src ? deg_norm(indptr, u0) : 1.0f;
        const float ns1 = scale_src ? deg_norm(indptr, u1) : 1.0f;
        const float4 *urow0 = reinterpret_cast<const float4 *>(x) + (base + u0) * c4;
        const float4 *urow1 = reinterpret_cast<const float4 *>(x) + (base + u1) * c4;
        float4 t0[4], t1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t q = q0 + lane + 64 * c;
          t0[c] = (q < c4) ? urow0[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          t1[c] = (q < c4) ? urow1[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          t0[c] = act(t0[c], c);
          t1[c] = act(t1[c], c);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // DGL scales the source features first, then sums (appendix A2)
          acc[c].x += t0[c].x * ns0; acc[c].y += t0[c].y * ns0; acc[c].z += t0[c].z * ns0; acc[c].w += t0[c].w * ns0;
          acc[c].x += t1[c].x * ns1; acc[c].y += t1[c].y * ns1; acc[c].z += t1[c].z * ns1; acc[c].w += t1[c].w * ns1;
        }
      }
      if (p < p1) {
        const int64_t u = indices[p];
        const float ns = scale_src ? deg_norm(indptr, u) : 1.0f;
        const float4 *urow = reinterpret_cast<const float4 *>(x) + (base + u) * c4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t q = q0 + lane + 64 * c;
          if (q < c4) {
            const float4 t = act(urow[q], c);
            acc[c].x += t.x * ns; acc[c].y += t.y * ns; acc[c].z += t.z * ns; acc[c].w += t.w * ns;
          }
        }
      }
      float4 *orow = reinterpret_cast<float4 *>(out) + r * c4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t q = q0 + lane + 64 * c;
        if (q < c4) {
          const float4 o = make_float4(acc[c].x * nd, acc[c].y * nd, acc[c].z * nd, acc[c].w * nd);
          orow[q] = o;
          omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
      }
    }
  }
  // partial maxima of |out| for the fp16 GEMM forms that read these rows next (tv_gemm_f32_ex,
  // state 2): an integer atomic max on the bits of the non-negative float, order-free
  if (out_max) {
    omax = tv_wave_max(omax);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(out_max) + (blockIdx.x & 255), __float_as_uint(omax));
  }
}

// thread per output element (narrow rows, e.g. the 2-column latency head)
__global__ __launch_bounds__(256) void graph_sum_scalar_kernel(
    int64_t rows_per_rep, int64_t rows, int64_t C, const int64_t *__restrict__ indptr,
    const int64_t *__restrict__ indices, const float *__restrict__ x, int scale_src,
    int scale_dst, float *__restrict__ out) {
  const int64_t total = rows * C, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / C, c = e - r * C;
    const int64_t rep = r / rows_per_rep, v = r - rep * rows_per_rep, base = rep * rows_per_rep;
    float acc = 0.f;
    for (int64_t p = indptr[v]; p < indptr[v + 1]; ++p) {
      const int64_t u = indices[p];
      const float ns = scale_src ? deg_norm(indptr, u) : 1.0f;
      acc += x[(base + u) * C + c] * ns;
    }
    out[e] = acc * (scale_dst ? deg_norm(indptr, v) : 1.0f);
  }
}

__global__ __launch_bounds__(256) void graph_norm_kernel(int64_t rows_per_rep, int64_t rows,
                                                         const int64_t *__restrict__ indptr,
                                                         float *__restrict__ norm) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) norm[r] = deg_norm(indptr, r % rows_per_rep);
}

// block (b, column tile): out[b,c] = mean over the rows of trace b
__global__ __launch_bounds__(256) void segment_mean_fwd_kernel(int64_t C,
                                                               const int64_t *__restrict__ off,
                                                               const float *__restrict__ x,
                                                               float *__restrict__ out) {
  const int64_t b = blockIdx.x;
  const int64_t c = (int64_t)blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const int64_t r0 = off[b], r1 = off[b + 1];
  float acc = 0.f;
  for (int64_t r = r0; r < r1; ++r) acc += x[r * C + c];
  out[b * C + c] = acc / (float)(r1 - r0);
}

// the same on float4 columns (C % 4 == 0, 16-byte aligned): four row loads in flight, rows
// still added in order
__global__ __launch_bounds__(256) void segment_mean_fwd_vec_kernel(int64_t C,
                                                                   const int64_t *__restrict__ off,
                                                                   const float *__restrict__ x,
                                                                   float *__restrict__ out,
                                                                   const float *__restrict__ act_scale,
                                                                   const float *__restrict__ act_shift,
                                                                   float act_slope) {
  const int64_t b = blockIdx.x, c4 = C >> 2;
  const int64_t q = (int64_t)blockIdx.y * 256 + threadIdx.x;
  if (q >= c4) return;
  const int64_t r0 = off[b], r1 = off[b + 1];
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  // optional prologue lrelu(v * scale[col] + shift[col]) on the rows as they are read (see
  // graph_sum_vec_kernel)
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act_scale) {
    sc = reinterpret_cast<const float4 *>(act_scale)[q];
    sh = reinterpret_cast<const float4 *>(act_shift)[q];
  }
  auto act = [&](float4 v) -> float4 {
    if (!act_scale) return v;
    return make_float4(tv_lrelu(fmaf(v.x, sc.x, sh.x), act_slope), tv_lrelu(fmaf(v.y, sc.y, sh.y), act_slope),
                       tv_lrelu(fmaf(v.z, sc.z, sh.z), act_slope), tv_lrelu(fmaf(v.w, sc.w, sh.w), act_slope));
  };
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t r = r0;
  for (; r + 4 <= r1; r += 4) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = act(x4[(r + j) * c4 + q]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
    }
  }
  for (; r < r1; ++r) {
    const float4 v = act(x4[r * c4 + q]);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float n = (float)(r1 - r0);
  reinterpret_cast<float4 *>(out)[b * c4 + q] = make_float4(acc.x / n, acc.y / n, acc.z / n, acc.w / n);
}

__global__ __launch_bounds__(256) void segment_mean_bwd_kernel(int64_t rows, int64_t C,
                                                               const int64_t *__restrict__ off,
                                                               const int64_t *__restrict__ gid,
                                                               const float *__restrict__ dout,
                                                               float *__restrict__ dx) {
  const int64_t total = rows * C, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / C, c = e - r * C;
    const int64_t g = gid[r];
    dx[e] = dout[g * C + c] / (float)(off[g + 1] - off[g]);
  }
}

// float4 columns: a thread owns a column chunk and walks rows (same division per element)
__global__ __launch_bounds__(256) void segment_mean_bwd_vec_kernel(int64_t rows, int64_t C,
                                                                   const int64_t *__restrict__ off,
                                                                   const int64_t *__restrict__ gid,
                                                                   const float *__restrict__ dout,
                                                                   float *__restrict__ dx) {
  const int64_t c4 = C >> 2;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= c4) return;
#pragma unroll 4
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const int64_t g = gid[r];
    const float n = (float)(off[g + 1] - off[g]);
    const float4 v = reinterpret_cast<const float4 *>(dout)[g * c4 + q];
    reinterpret_cast<float4 *>(dx)[r * c4 + q] = make_float4(v.x / n, v.y / n, v.z / n, v.w / n);
  }
}

__global__ __launch_bounds__(256) void embedding_fwd_kernel(int64_t rows, int64_t dim, int64_t V,
                                                            const float *__restrict__ w,
                                                            const int64_t *__restrict__ ids,
                                                            float *__restrict__ out, int64_t ld,
                                                            int64_t col0) {
  const int64_t total = rows * dim, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / dim, d = e - r * dim;
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    out[r * ld + col0 + d] = w[id * dim + d];
  }
}

// dW[v,:] = sum over rows with ids[r] == v.  One workgroup per EMB_VB consecutive operation
// ids; its 4 waves scan the id list in 64-row chunks (chunk c belongs to wave c % 4), pick the
// matching rows with a ballot and add them, in row order, into the wave's LDS partial; the four
// partials are combined in wave order.  Fixed summation order, no atomics; every dout row is
// read by exactly one workgroup.
constexpr int EMB_VB = 4;
__global__ __launch_bounds__(256) void embedding_bwd_kernel(int64_t rows, int64_t dim, int64_t V,
                                                            const int64_t *__restrict__ ids,
                                                            const float *__restrict__ dout,
                                                            int64_t ld, int64_t col0,
                                                            float *__restrict__ dw) {
  __shared__ float part[4][EMB_VB][64];
  const int64_t v0 = (int64_t)blockIdx.x * EMB_VB;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int64_t d0 = 0; d0 < dim; d0 += 64) {
    const int64_t d = d0 + lane;
#pragma unroll
    for (int j = 0; j < EMB_VB; ++j) part[grp][j][lane] = 0.f;
    for (int64_t r0 = (int64_t)grp * 64; r0 < rows; r0 += 256) {
      const int64_t r = r0 + lane;
      int64_t id = r < rows ? ids[r] : -1;
      id = (r < rows) ? (id < 0 ? 0 : (id >= V ? V - 1 : id)) : -1;   // clamp like the forward
      const int rel = (int)(id - v0);
      unsigned long long hits = __ballot(rel >= 0 && rel < EMB_VB);
      while (hits) {
        const int bit = __builtin_ctzll(hits);
        hits &= hits - 1;
        const int j = __shfl(rel, bit);
        if (d < dim) part[grp][j][lane] += dout[(r0 + bit) * ld + col0 + d];
      }
    }
    __syncthreads();
    if (grp == 0 && d < dim) {
#pragma unroll
      for (int j = 0; j < EMB_VB; ++j)
        if (v0 + j < V)
          dw[(v0 + j) * dim + d] = ((part[0][j][lane] + part[1][j][lane]) + part[2][j][lane]) + part[3][j][lane];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void latdec_input_fwd_kernel(
    int64_t K, int64_t B, int64_t M, int64_t Z, int64_t E, int64_t V,
    const float *__restrict__ z2, const float *__restrict__ emb,
    const int64_t *__restrict__ ntp, float *__restrict__ out) {
  const int64_t W = Z + E, total = K * B * M * W, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t row = e / W, c = e - row * W;
    const int64_t kb = row / M, m = row - kb * M, b = kb % B;
    if (c < Z) {
      out[e] = z2[kb * Z + c];
    } else {
      int64_t id = ntp[b * M + m];
      id = id < 0 ? 0 : (id >= V ? V - 1 : id);
      out[e] = emb[id * E + (c - Z)];
    }
  }
}

__global__ __launch_bounds__(256) void latdec_input_bwd_kernel(int64_t KB, int64_t M, int64_t Z,
                                                               int64_t W,
                                                               const float *__restrict__ dout,
                                                               float *__restrict__ dz2) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= KB * Z) return;
  const int64_t kb = e / Z, d = e - kb * Z;
  float acc = 0.f;
  for (int64_t m = 0; m < M; ++m) acc += dout[(kb * M + m) * W + d];
  dz2[e] = acc;
}

// compact latency-decoder input: rows of one replica = [real nodes (N) | one representative
// of the padded slots per trace (B)]; row i belongs to trace row_graph[i] and has operation
// row_type[i]:  out[k*R + i] = [z2[k, row_graph[i]] || emb[row_type[i]]]
__global__ __launch_bounds__(256) void latdec_rows_fwd_kernel(
    int64_t K, int64_t R, int64_t B, int64_t Z, int64_t E, int64_t V,
    const float *__restrict__ z2, const float *__restrict__ emb,
    const int64_t *__restrict__ row_graph, const int64_t *__restrict__ row_type,
    float *__restrict__ out) {
  const int64_t W = Z + E, total = K * R * W, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t row = e / W, c = e - row * W;
    const int64_t k = row / R, i = row - k * R;
    if (c < Z) {
      out[e] = z2[(k * B + row_graph[i]) * Z + c];
    } else {
      int64_t id = row_type[i];
      id = id < 0 ? 0 : (id >= V ? V - 1 : id);
      out[e] = emb[id * E + (c - Z)];
    }
  }
}

// dz2[k,b,:] = sum of dout[:, :Z] over the rows of trace b, node order, representative last
__global__ __launch_bounds__(256) void latdec_rows_bwd_kernel(
    int64_t K, int64_t R, int64_t B, int64_t Z, int64_t W,
    const int64_t *__restrict__ off, const float *__restrict__ dout, float *__restrict__ dz2) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * B * Z) return;
  const int64_t kb = e / Z, d = e - kb * Z;
  const int64_t k = kb / B, b = kb - k * B;
  const int64_t N = off[B];
  float acc = 0.f;
  for (int64_t i = off[b]; i < off[b + 1]; ++i) acc += dout[(k * R + i) * W + d];
  acc += dout[(k * R + N + b) * W + d];
  dz2[e] = acc;
}

inline unsigned ew_blocks(int64_t work_items) {
  int64_t b = tv_cdiv(work_items, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int tv_graph_sum(int64_t rows_per_rep, int64_t n_rep, int64_t cols,
                            const int64_t *indptr, const int64_t *indices, const float *x,
                            int scale_src, int scale_dst, float *out, float *out_max,
                            const float *act_scale, const float *act_shift, float act_slope,
                            void *stream) {
  TV_REQUIRE(rows_per_rep >= 0 && n_rep >= 1 && cols >= 1, "tv_graph_sum: bad sizes");
  const int64_t rows = rows_per_rep * n_rep;
  if (rows == 0) return TV_OK;
  TV_REQUIRE(indptr && indices && x && out, "tv_graph_sum: null pointer");
  TV_REQUIRE(x != out, "tv_graph_sum: in-place aggregation is not possible");
  hipStream_t s = (hipStream_t)stream;
  const bool vec = cols % 4 == 0 && cols >= 64 && ((((uintptr_t)x | (uintptr_t)out) & 15) == 0);
  TV_REQUIRE(!out_max || vec, "tv_graph_sum: out_max needs rows of a multiple of 4 floats (>= 64), 16-byte aligned");
  TV_REQUIRE((act_scale == nullptr) == (act_shift == nullptr), "tv_graph_sum: act_scale and act_shift go together");
  TV_REQUIRE(!act_scale || (vec && ((((uintptr_t)act_scale | (uintptr_t)act_shift) & 15) == 0)),
             "tv_graph_sum: the prologue needs rows of a multiple of 4 floats (>= 64) and 16-byte aligned operands");
  if (vec) {
    int64_t blocks = tv_cdiv(rows, 4);
    if (blocks > 8192) blocks = 8192;
    if (out_max) TV_HIP(hipMemsetAsync(out_max, 0, 256 * sizeof(float), s));
    hipLaunchKernelGGL(graph_sum_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rows_per_rep,
                       rows, cols / 4, indptr, indices, x, scale_src, scale_dst, out, out_max, act_scale,
                       act_shift, act_slope);
  } else {
    hipLaunchKernelGGL(graph_sum_scalar_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, s,
                       rows_per_rep, rows, cols, indptr, indices, x, scale_src, scale_dst, out);
  }
  TV_LAUNCH_CHECK("tv_graph_sum");
  return TV_OK;
}

extern "C" int tv_graph_norm(int64_t rows_per_rep, int64_t n_rep, const int64_t *indptr,
                             float *norm, void *stream) {
  TV_REQUIRE(rows_per_rep >= 0 && n_rep >= 1, "tv_graph_norm: bad sizes");
  const int64_t rows = rows_per_rep * n_rep;
  if (rows == 0) return TV_OK;
  TV_REQUIRE(indptr && norm, "tv_graph_norm: null pointer");
  hipLaunchKernelGGL(graph_norm_kernel, dim3((unsigned)tv_cdiv(rows, 256)), dim3(256), 0,
                     (hipStream_t)stream, rows_per_rep, rows, indptr, norm);
  TV_LAUNCH_CHECK("tv_graph_norm");
  return TV_OK;
}

extern "C" int tv_segment_mean_fwd(int64_t batch, int64_t cols, const int64_t *node_offsets,
                                   const float *x, float *out, const float *act_scale,
                                   const float *act_shift, float act_slope, void *stream) {
  TV_REQUIRE(batch >= 0 && cols >= 1, "tv_segment_mean_fwd: bad sizes");
  if (batch == 0) return TV_OK;
  TV_REQUIRE(node_offsets && x && out, "tv_segment_mean_fwd: null pointer");
  const bool vec = cols % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
  TV_REQUIRE((act_scale == nullptr) == (act_shift == nullptr),
             "tv_segment_mean_fwd: act_scale and act_shift go together");
  TV_REQUIRE(!act_scale || (vec && ((((uintptr_t)act_scale | (uintptr_t)act_shift) & 15) == 0)),
             "tv_segment_mean_fwd: the prologue needs rows of a multiple of 4 floats and 16-byte aligned operands");
  if (vec) {
    const dim3 grid((unsigned)batch, (unsigned)tv_cdiv(cols / 4, 256));
    hipLaunchKernelGGL(segment_mean_fwd_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, cols,
                       node_offsets, x, out, act_scale, act_shift, act_slope);
  } else {
    const dim3 grid((unsigned)batch, (unsigned)tv_cdiv(cols, 256));
    hipLaunchKernelGGL(segment_mean_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, cols,
                       node_offsets, x, out);
  }
  TV_LAUNCH_CHECK("tv_segment_mean_fwd");
  return TV_OK;
}

extern "C" int tv_segment_mean_bwd(int64_t rows, int64_t cols, const int64_t *node_offsets,
                                   const int64_t *graph_id, const float *dout, float *dx,
                                   void *stream) {
  TV_REQUIRE(rows >= 0 && cols >= 1, "tv_segment_mean_bwd: bad sizes");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(node_offsets && graph_id && dout && dx, "tv_segment_mean_bwd: null pointer");
  if (cols % 4 == 0 && (((uintptr_t)dout | (uintptr_t)dx) & 15) == 0) {
    const int64_t cblocks = tv_cdiv(cols / 4, 256);
    int64_t rblocks = tv_cdiv(4096, cblocks);
    if (rblocks > rows) rblocks = rows;
    hipLaunchKernelGGL(segment_mean_bwd_vec_kernel, dim3((unsigned)cblocks, (unsigned)rblocks),
                       dim3(256), 0, (hipStream_t)stream, rows, cols, node_offsets, graph_id, dout, dx);
  } else {
    hipLaunchKernelGGL(segment_mean_bwd_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0,
                       (hipStream_t)stream, rows, cols, node_offsets, graph_id, dout, dx);
  }
  TV_LAUNCH_CHECK("tv_segment_mean_bwd");
  return TV_OK;
}

extern "C" int tv_embedding_fwd(int64_t rows, int64_t dim, int64_t num_operations,
                                const float *weight, const int64_t *ids, float *out, int64_t ld,
                                int64_t col0, void *stream) {
  TV_REQUIRE(rows >= 0 && dim >= 1 && num_operations >= 1, "tv_embedding_fwd: bad sizes");
  TV_REQUIRE(col0 >= 0 && ld >= col0 + dim, "tv_embedding_fwd: columns do not fit ld");
  if (rows == 0) return TV_OK;
  TV_REQUIRE(weight && ids && out, "tv_embedding_fwd: null pointer");
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_blocks(rows * dim)), dim3(256), 0,
                     (hipStream_t)stream, rows, dim, num_operations, weight, ids, out, ld, col0);
  TV_LAUNCH_CHECK("tv_embedding_fwd");
  return TV_OK;
}

extern "C" int tv_embedding_bwd(int64_t rows, int64_t dim, int64_t num_operations,
                                const int64_t *ids, const float *dout, int64_t ld, int64_t col0,
                                float *dweight, void *stream) {
  TV_REQUIRE(rows >= 0 && dim >= 1 && num_operations >= 1, "tv_embedding_bwd: bad sizes");
  TV_REQUIRE(col0 >= 0 && ld >= col0 + dim, "tv_embedding_bwd: columns do not fit ld");
  TV_REQUIRE(ids && dout && dweight, "tv_embedding_bwd: null pointer");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)tv_cdiv(num_operations, EMB_VB)), dim3(256),
                     0, (hipStream_t)stream, rows, dim, num_operations, ids, dout, ld, col0, dweight);
  TV_LAUNCH_CHECK("tv_embedding_bwd");
  return TV_OK;
}

extern "C" int tv_latdec_input_fwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                   int64_t z2_dim, int64_t emb_dim, int64_t num_operations,
                                   const float *z2, const float *emb_weight,
                                   const int64_t *node_type_pad, float *out, void *stream) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0 && max_node_count >= 1 && z2_dim >= 1 && emb_dim >= 1 &&
                 num_operations >= 1,
             "tv_latdec_input_fwd: bad sizes");
  if (batch == 0) return TV_OK;
  TV_REQUIRE(z2 && emb_weight && node_type_pad && out, "tv_latdec_input_fwd: null pointer");
  const int64_t total = n_samples * batch * max_node_count * (z2_dim + emb_dim);
  hipLaunchKernelGGL(latdec_input_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0,
                     (hipStream_t)stream, n_samples, batch, max_node_count, z2_dim, emb_dim,
                     num_operations, z2, emb_weight, node_type_pad, out);
  TV_LAUNCH_CHECK("tv_latdec_input_fwd");
  return TV_OK;
}

extern "C" int tv_latdec_input_bwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                   int64_t z2_dim, int64_t emb_dim, const float *dout, float *dz2,
                                   void *stream) {
  TV_REQUIRE(n_samples >= 1 && batch >= 0 && max_node_count >= 1 && z2_dim >= 1 && emb_dim >= 0,
             "tv_latdec_input_bwd: bad sizes");
  if (batch == 0) return TV_OK;
  TV_REQUIRE(dout && dz2, "tv_latdec_input_bwd: null pointer");
  const int64_t kb = n_samples * batch;
  hipLaunchKernelGGL(latdec_input_bwd_kernel, dim3((unsigned)tv_cdiv(kb * z2_dim, 256)), dim3(256),
                     0, (hipStream_t)stream, kb, max_node_count, z2_dim, z2_dim + emb_dim, dout,
                     dz2);
  TV_LAUNCH_CHECK("tv_latdec_input_bwd");
  return TV_OK;
}

extern "C" int tv_latdec_rows_fwd(int64_t n_samples, int64_t rows_per_rep, int64_t batch,
                                  int64_t z2_dim, int64_t emb_dim, int64_t num_operations,
                                  const float *z2, const float *emb_weight,
                                  const int64_t *row_graph, const int64_t *row_type, float *out,
                                  void *stream) {
  TV_REQUIRE(n_samples >= 1 && rows_per_rep >= 0 && batch >= 0 && z2_dim >= 1 && emb_dim >= 1 &&
                 num_operations >= 1,
             "tv_latdec_rows_fwd: bad sizes");
  if (rows_per_rep == 0) return TV_OK;
  TV_REQUIRE(z2 && emb_weight && row_graph && row_type && out, "tv_latdec_rows_fwd: null pointer");
  const int64_t total = n_samples * rows_per_rep * (z2_dim + emb_dim);
  hipLaunchKernelGGL(latdec_rows_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0,
                     (hipStream_t)stream, n_samples, rows_per_rep, batch, z2_dim, emb_dim,
                     num_operations, z2, emb_weight, row_graph, row_type, out);
  TV_LAUNCH_CHECK("tv_latdec_rows_fwd");
  return TV_OK;
}

extern "C" int tv_latdec_rows_bwd(int64_t n_samples, int64_t rows_per_rep, int64_t batch,
                                  int64_t z2_dim, int64_t emb_dim, const int64_t *node_offsets,
                                  const float *dout, float *dz2, void *stream) {
  TV_REQUIRE(n_samples >= 1 && rows_per_rep >= batch && batch >= 0 && z2_dim >= 1 && emb_dim >= 0,
             "tv_latdec_rows_bwd: bad sizes");
  if (batch == 0) return TV_OK;
  TV_REQUIRE(node_offsets && dout && dz2, "tv_latdec_rows_bwd: null pointer");
  hipLaunchKernelGGL(latdec_rows_bwd_kernel,
                     dim3((unsigned)tv_cdiv(n_samples * batch * z2_dim, 256)), dim3(256), 0,
                     (hipStream_t)stream, n_samples, rows_per_rep, batch, z2_dim,
                     z2_dim + emb_dim, node_offsets, dout, dz2);
  TV_LAUNCH_CHECK("tv_latdec_rows_bwd");
  return TV_OK;
}
