// (f4) decoder graphs from a SAMPLED adjacency: what model/struct_vae.py:296-333 (make_graph)
// does per sample with T.where + dgl.graph + add_reverse_edges + add_self_loop + dgl.batch,
// as index kernels.  One workgroup per graph; HBM-bound integer work:
//   read 4*T (adjacency) twice, write 8*(5*E + 2*M) per graph, plus 4*T of rank scratch.
//
// Reference behaviour restated (G graphs of M nodes, node ids g*M + local):
//   (u, v) = where(triu_to_dense(adj))           -> forward edges in row-major order, u < v
//   add_reverse_edges                            -> the same list with (v, u), appended
//   add_self_loop                                -> (x, x) for x < M, appended
//   node_idx[v] = max(max_u(adj[u,v] * cumsum_v(adj[u,:])) - 1, 0)
//   CSC = stable sort by destination: in-edges of x in edge-id order
#include "tv_common.h"

namespace {

constexpr int SG_THREADS = 256;

__device__ __forceinline__ int64_t tri(int64_t M, int64_t u, int64_t w) {
  // tensor_utils.py:60-67: row-major strict upper triangle
  return u * (2 * M - u - 1) / 2 + (w - u - 1);
}

// The per-node walks below read a graph's adjacency M times over, one dependent load per
// trip: with STAGED the workgroup first copies it into LDS as bytes (one coalesced pass), which
// is what fits for M up to ~350; longer traces read it from global memory (L2) instead.
template <bool STAGED>
struct Adjacency {
  const float *A;
  const uint8_t *S;
  __device__ __forceinline__ bool operator()(int64_t t) const {
    return STAGED ? S[t] != 0 : A[t] != 0.f;
  }
};

// LDS: int32 ints[n_ints], then (STAGED) T bytes.  Returns true if a value is neither 0 nor 1.
template <bool STAGED>
__device__ __forceinline__ bool stage_adjacency(const float *A, int64_t T, uint8_t *S,
                                                bool validate) {
  bool bad = false;
  if (STAGED || validate) {
    for (int64_t t = threadIdx.x; t < T; t += SG_THREADS) {
      const float x = A[t];
      bad |= (x != 0.f && x != 1.f);
      if (STAGED) S[t] = (x != 0.f);
    }
  }
  return bad;
}

// degrees -> graph-local exclusive prefix sums (into indptr) and the graph's edge total
template <bool STAGED>
__global__ __launch_bounds__(SG_THREADS) void sg_count_kernel(int64_t M, const float *adj,
                                                              int64_t *indptr, int64_t *totals,
                                                              int32_t *status) {
  extern __shared__ int32_t lds[];
  const int64_t g = blockIdx.x;
  const int64_t T = M * (M - 1) / 2;
  const float *A = adj + g * T;
  uint8_t *S = reinterpret_cast<uint8_t *>(lds + (M + 1));
  const bool bad = stage_adjacency<STAGED>(A, T, S, true);
  if (bad) atomicOr(status, TV_SAMPLED_BAD_ADJ);
  if (STAGED) __syncthreads();
  const Adjacency<STAGED> nz{A, S};
  for (int64_t v = threadIdx.x; v < M; v += SG_THREADS) {
    int deg = 1;
    for (int64_t u = 0; u < v; ++u) deg += nz(tri(M, u, v));
    for (int64_t w = v + 1; w < M; ++w) deg += nz(tri(M, v, w));
    lds[v] = deg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int64_t v = 0; v < M; ++v) {
      const int d = lds[v];
      lds[v] = run;
      run += d;
    }
    lds[M] = run;
  }
  __syncthreads();
  for (int64_t v = threadIdx.x; v < M; v += SG_THREADS) indptr[g * M + v] = lds[v];
  if (threadIdx.x == 0) totals[g] = lds[M];
}

// exclusive scan of the G totals, one block: offs[g] = sum_{g' < g} totals[g'], offs[G] = E
__global__ __launch_bounds__(SG_THREADS) void sg_scan_kernel(int64_t G, const int64_t *totals,
                                                             int64_t *offs) {
  __shared__ int64_t part[SG_THREADS + 1];
  const int64_t chunk = (G + SG_THREADS - 1) / SG_THREADS;
  const int64_t lo = (int64_t)threadIdx.x * chunk;
  const int64_t hi = lo + chunk < G ? lo + chunk : G;
  int64_t sum = 0;
  for (int64_t g = lo; g < hi; ++g) sum += totals[g];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int i = 0; i < SG_THREADS; ++i) {
      const int64_t s = part[i];
      part[i] = run;
      run += s;
    }
    part[SG_THREADS] = run;
  }
  __syncthreads();
  int64_t run = part[threadIdx.x];
  for (int64_t g = lo; g < hi; ++g) {
    offs[g] = run;
    run += totals[g];
  }
  if (threadIdx.x == 0) offs[G] = part[SG_THREADS];
}

__global__ __launch_bounds__(SG_THREADS) void sg_offset_kernel(int64_t G, int64_t M,
                                                               const int64_t *offs,
                                                               int64_t *indptr) {
  const int64_t g = blockIdx.x;
  const int64_t base = offs[g];
  for (int64_t v = threadIdx.x; v < M; v += SG_THREADS) indptr[g * M + v] += base;
  if (g == G - 1 && threadIdx.x == 0) indptr[G * M] = offs[G];
}

struct FillArgs {
  int64_t G, M, E;
  const float *adj;
  const int64_t *offs, *indptr;
  int64_t *src, *dst, *indices, *eids, *rev, *node_idx;
  int32_t *rank;   // [G, T] scratch: position of an edge among the nonzeros of its row
};

// LDS (int32): lptr[M] colcnt[M] rowcnt[M] rowstart[M+1], then (STAGED) T bytes
template <bool STAGED>
__global__ __launch_bounds__(SG_THREADS) void sg_fill_kernel(FillArgs a) {
  extern __shared__ int32_t lds[];
  const int64_t g = blockIdx.x, M = a.M;
  const int64_t T = M * (M - 1) / 2;
  const float *A = a.adj + g * T;
  int32_t *R = a.rank + g * T;
  int32_t *lptr = lds, *colcnt = lptr + M, *rowcnt = colcnt + M, *rowstart = rowcnt + M;
  uint8_t *S = reinterpret_cast<uint8_t *>(rowstart + (M + 1));
  const int64_t E0 = a.offs[g];          // first edge (and first CSC position) of this graph
  const int64_t node0 = g * M;
  stage_adjacency<STAGED>(A, T, S, false);
  if (STAGED) __syncthreads();
  const Adjacency<STAGED> nz{A, S};

  for (int64_t v = threadIdx.x; v < M; v += SG_THREADS) {
    int c = 0, r = 0;
    for (int64_t u = 0; u < v; ++u) c += nz(tri(M, u, v));
    for (int64_t w = v + 1; w < M; ++w) r += nz(tri(M, v, w));
    colcnt[v] = c;
    rowcnt[v] = r;
    lptr[v] = (int32_t)(a.indptr[node0 + v] - E0);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int64_t v = 0; v < M; ++v) {
      rowstart[v] = run;
      run += rowcnt[v];
    }
    rowstart[M] = run;
  }
  __syncthreads();
  const int64_t nnz = rowstart[M];
  // the arrays were sized from tv_sampled_graph_count's result: an adjacency that no longer
  // matches it must not be written out (uniform over the workgroup)
  if (a.offs[g + 1] - E0 != M + 2 * nnz || a.offs[g + 1] > a.E || E0 < 0) return;

  // rows: COO of the forward and reverse edges, CSC entries of the reverse edges and the
  // self loop (they follow the forward in-edges of the row's node), rank scratch
  for (int64_t u = threadIdx.x; u < M; u += SG_THREADS) {
    const int64_t pos0 = E0 + lptr[u] + colcnt[u];
    int k = 0;
    for (int64_t w = u + 1; w < M; ++w) {
      const int64_t t = tri(M, u, w);
      if (nz(t)) {
        const int64_t e = rowstart[u] + k;
        R[t] = k;
        a.src[E0 + e] = node0 + u;
        a.dst[E0 + e] = node0 + w;
        a.src[E0 + nnz + e] = node0 + w;
        a.dst[E0 + nnz + e] = node0 + u;
        a.indices[pos0 + k] = node0 + w;
        a.eids[pos0 + k] = E0 + nnz + e;
        ++k;
      }
    }
    const int64_t ps = pos0 + k, es = E0 + 2 * nnz + u;
    a.src[es] = node0 + u;
    a.dst[es] = node0 + u;
    a.indices[ps] = node0 + u;
    a.eids[ps] = es;
    a.rev[ps] = ps;
  }
  // the rank scratch is read by other threads of this workgroup below
  __threadfence();
  __syncthreads();

  // columns: CSC entries of the forward edges, the reverse links, node_idx
  for (int64_t w = threadIdx.x; w < M; w += SG_THREADS) {
    int c = 0, best = 0;
    for (int64_t u = 0; u < w; ++u) {
      const int64_t t = tri(M, u, w);
      if (nz(t)) {
        const int k = R[t];
        const int64_t pf = E0 + lptr[w] + c;
        const int64_t pr = E0 + lptr[u] + colcnt[u] + k;
        a.indices[pf] = node0 + u;
        a.eids[pf] = E0 + rowstart[u] + k;
        a.rev[pf] = pr;
        a.rev[pr] = pf;
        best = k > best ? k : best;
        ++c;
      }
    }
    a.node_idx[node0 + w] = best;
  }
}

// the byte copy of one adjacency next to the int32 arrays, within the 64 KB a launch gets
// without asking for more
inline bool staged_fits(size_t int_bytes, int64_t M) {
  return int_bytes + (size_t)(M * (M - 1) / 2) <= 64 * 1024;
}

}  // namespace

extern "C" int64_t tv_sampled_graph_workspace_bytes(int64_t graphs, int64_t max_node_count) {
  if (graphs < 0 || max_node_count < 2) return -1;
  const int64_t T = max_node_count * (max_node_count - 1) / 2;
  // [G] totals (int64) + [G, T] ranks (int32)
  return 8 * (graphs + 1) + 4 * graphs * T;
}

extern "C" int tv_sampled_graph_count(int64_t graphs, int64_t max_node_count, const float *adj,
                                      int64_t *indptr, int64_t *graph_edge_offsets, void *ws,
                                      int64_t ws_bytes, int32_t *status, void *stream) {
  TV_REQUIRE(graphs >= 0, "tv_sampled_graph_count: graphs must be >= 0");
  TV_REQUIRE(max_node_count >= 2 && max_node_count <= 1536,
             "tv_sampled_graph_count: max_node_count must be in [2, 1536]: got %lld",
             (long long)max_node_count);
  TV_REQUIRE(graphs < (1ll << 31), "tv_sampled_graph_count: too many graphs");
  TV_REQUIRE(indptr && graph_edge_offsets && status, "tv_sampled_graph_count: null output");
  hipStream_t s = (hipStream_t)stream;
  if (graphs == 0) {
    TV_HIP(hipMemsetAsync(indptr, 0, sizeof(int64_t), s));
    TV_HIP(hipMemsetAsync(graph_edge_offsets, 0, sizeof(int64_t), s));
    return TV_OK;
  }
  TV_REQUIRE(adj && ws, "tv_sampled_graph_count: null input");
  if (ws_bytes < tv_sampled_graph_workspace_bytes(graphs, max_node_count)) {
    tv_set_error("tv_sampled_graph_count: workspace of %lld bytes is too small",
                 (long long)ws_bytes);
    return TV_ERR_WORKSPACE;
  }
  int64_t *totals = (int64_t *)ws;
  const int64_t M = max_node_count;
  const size_t ints = sizeof(int32_t) * (size_t)(M + 1);
  if (staged_fits(ints, M))
    hipLaunchKernelGGL(sg_count_kernel<true>, dim3((unsigned)graphs), dim3(SG_THREADS),
                       ints + (size_t)(M * (M - 1) / 2), s, M, adj, indptr, totals, status);
  else
    hipLaunchKernelGGL(sg_count_kernel<false>, dim3((unsigned)graphs), dim3(SG_THREADS), ints, s,
                       M, adj, indptr, totals, status);
  TV_LAUNCH_CHECK("tv_sampled_graph_count");
  hipLaunchKernelGGL(sg_scan_kernel, dim3(1), dim3(SG_THREADS), 0, s, graphs, totals,
                     graph_edge_offsets);
  TV_LAUNCH_CHECK("tv_sampled_graph_count (scan)");
  hipLaunchKernelGGL(sg_offset_kernel, dim3((unsigned)graphs), dim3(SG_THREADS), 0, s, graphs, M,
                     graph_edge_offsets, indptr);
  TV_LAUNCH_CHECK("tv_sampled_graph_count (offsets)");
  return TV_OK;
}

extern "C" int tv_sampled_graph_fill(int64_t graphs, int64_t max_node_count, int64_t num_edges,
                                     const float *adj, const int64_t *graph_edge_offsets,
                                     const int64_t *indptr, int64_t *src, int64_t *dst,
                                     int64_t *indices, int64_t *eids, int64_t *rev,
                                     int64_t *node_idx, void *ws, int64_t ws_bytes,
                                     void *stream) {
  TV_REQUIRE(graphs >= 0 && graphs < (1ll << 31), "tv_sampled_graph_fill: bad graph count");
  TV_REQUIRE(max_node_count >= 2 && max_node_count <= 1536,
             "tv_sampled_graph_fill: max_node_count must be in [2, 1536]: got %lld",
             (long long)max_node_count);
  if (graphs == 0) return TV_OK;
  const int64_t M = max_node_count, T = M * (M - 1) / 2;
  // every graph has its M self loops and at most 2T more edges
  TV_REQUIRE(num_edges >= graphs * M && num_edges <= graphs * (M + 2 * T),
             "tv_sampled_graph_fill: num_edges %lld is impossible for %lld graphs of %lld nodes",
             (long long)num_edges, (long long)graphs, (long long)M);
  TV_REQUIRE(adj && graph_edge_offsets && indptr && ws, "tv_sampled_graph_fill: null input");
  TV_REQUIRE(src && dst && indices && eids && rev && node_idx,
             "tv_sampled_graph_fill: null output");
  if (ws_bytes < tv_sampled_graph_workspace_bytes(graphs, max_node_count)) {
    tv_set_error("tv_sampled_graph_fill: workspace of %lld bytes is too small",
                 (long long)ws_bytes);
    return TV_ERR_WORKSPACE;
  }
  FillArgs a;
  a.G = graphs; a.M = M; a.E = num_edges; a.adj = adj; a.offs = graph_edge_offsets; a.indptr = indptr;
  a.src = src; a.dst = dst; a.indices = indices; a.eids = eids; a.rev = rev;
  a.node_idx = node_idx;
  a.rank = (int32_t *)((int64_t *)ws + (graphs + 1));
  const size_t ints = sizeof(int32_t) * (size_t)(4 * M + 1);
  if (staged_fits(ints, M))
    hipLaunchKernelGGL(sg_fill_kernel<true>, dim3((unsigned)graphs), dim3(SG_THREADS),
                       ints + (size_t)T, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(sg_fill_kernel<false>, dim3((unsigned)graphs), dim3(SG_THREADS), ints,
                       (hipStream_t)stream, a);
  TV_LAUNCH_CHECK("tv_sampled_graph_fill");
  return TV_OK;
}
