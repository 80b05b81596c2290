// (a) batching: trace store in HBM -> batched COO + destination-sorted CSC + observed
// tensors.  One workgroup per trace; the per-trace index work happens in LDS (per-edge
// parallel, O(n)), the outputs are written once, coalesced.  HBM-bound integer work: ~ (24 B in + 16*E_enc + 16*E_dec +
// 8*(N + B*M) + observed tensors) per batch.
//
// Reference behaviour restated (see include/tracevae_hip.h for the call sites):
//   encoder graph of trace i : edges [u->v ; v->u ; (x,x) x<n]   (dataset.py:39-52)
//   decoder graph of trace i : edges [u->v ; v->u ; (x,x) x<M]   (struct_vae.py:338-347)
//   CSC = stable sort by destination: in-edges of x in edge-id order
#include "tv_common.h"

namespace {

struct BatchArgs {
  int64_t B, M, N, V;
  const int64_t *node_offsets, *store_node_start, *store_edge_start;
  const int64_t *store_u, *store_v, *store_node_type;
  const float *store_avg, *store_min, *store_max;
  const double *lat_mean, *lat_std;
  int64_t *enc_src, *enc_dst, *enc_indptr, *enc_indices, *enc_eids, *enc_rev;
  int64_t *dec_src, *dec_dst, *dec_indptr, *dec_indices, *dec_eids, *dec_rev;
  float *adj;
  int64_t *node_type_pad;
  float *latency_pad;
  int64_t *node_count, *node_type;
  float *latency;
  int64_t *graph_id;
  float *lat_row_weight;   // optional: also emit the compact latency-decoder graph tails
  int32_t *status;
};

__device__ __forceinline__ float normalize_latency(float x, double mu, double sd) {
  // dataset.py:78: (latency - mu) / (std + 1e-5) in float64, then cast to float32
  return (float)(((double)x - mu) / (sd + 1e-5));
}

// One workgroup per trace (64 lanes for short traces, 256 for long ones); per-EDGE parallel
// index work in LDS, O(n) in all:
//   in-edges of node x in edge-id order = [e with v[e] == x ; e with u[e] == x ; self], so the CSC
//   position of edge e is   pA = indptr[v] + rankA(e)            (the entry "u -> v" in v's list)
//                           pB = indptr[u] + cntA[u] + rankB(e)  (the entry "v -> u" in u's list)
//   with cntA / cntB the per-node counts (integer LDS atomics: order-free sums) and rankA / rankB
//   the rank of e among the earlier edges with the same v / u.  The two entries are each other's
//   reverse: rev[pA] = pB, rev[pB] = pA - no search.
//   graph_vectors() emits edges in BFS order (trace_graph.py:286-352): v strictly increasing, u
//   non-decreasing; then rankA = 0 and rankB = e - (first edge of u).  Any other order of a valid
//   edge list takes the general ranks (a count over the earlier edges).
// LDS (int32): u[M] v[M] indptr[M+1] cntA[M] cntB[M] first[M] nbr[3M] eid[3M] rev[3M]
constexpr int BB_LDS_INTS_PER_NODE = 15;

__global__ __launch_bounds__(256) void batch_build_kernel(BatchArgs a) {
  extern __shared__ int32_t lds[];
  const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  const int64_t i = blockIdx.x;
  const int64_t M = a.M;
  int32_t *su = lds;
  int32_t *sv = su + M;
  int32_t *sptr = sv + M;
  int32_t *scA = sptr + (M + 1);
  int32_t *scB = scA + M;
  int32_t *sfirst = scB + M;
  int32_t *snbr = sfirst + M;
  int32_t *seid = snbr + 3 * M;
  int32_t *srev = seid + 3 * M;

  const int64_t n0 = a.node_offsets[i];
  const int64_t n64 = a.node_offsets[i + 1] - n0;
  const int64_t E0 = 3 * n0 - 2 * i;            // first encoder edge of this trace
  const int64_t D0 = i * M + 2 * (n0 - i);      // first decoder edge of this trace
  const int64_t dbase = i * M;                  // first decoder node of this trace
  const bool last = (i == a.B - 1);

  // Compact latency-decoder graph = encoder graph + one self-loop node per trace (row N + i)
  // standing for the trace's M - n padded slots: the caller sized the encoder arrays, node_type
  // and graph_id with B more entries.
  if (a.lat_row_weight && tid == 0) {
    const int64_t Et = 3 * a.N - 2 * a.B, p = Et + i, row = a.N + i;
    a.enc_src[p] = row; a.enc_dst[p] = row;
    a.enc_indices[p] = row; a.enc_eids[p] = p; a.enc_rev[p] = p;
    a.enc_indptr[row + 1] = p + 1;
    a.node_type[row] = 0;
    a.graph_id[row] = i;
    a.lat_row_weight[row] = (n64 >= 1 && n64 <= M) ? (float)(M - n64) : 0.f;
  }
  if (n64 < 1 || n64 > M) {
    if (tid == 0) atomicOr(a.status, TV_BATCH_TOO_MANY_NODES);
    // keep the decoder structure of this slot well-formed (self loops only)
    for (int64_t x = tid; x < M; x += nth) {
      a.dec_indptr[dbase + x] = D0 + x;
    }
    return;
  }
  const int n = (int)n64;
  const int ne = n - 1;
  const int64_t sn = a.store_node_start[i];
  const int64_t se = a.store_edge_start[i];

  // ---- edges into LDS + validation; per-node counters cleared
  int bad = 0;
  for (int e = tid; e < ne; e += nth) {
    int64_t u = a.store_u[se + e], v = a.store_v[se + e];
    if (!(u >= 0 && u < v && v < n)) { bad = 1; u = 0; v = 1; }
    su[e] = (int32_t)u;
    sv[e] = (int32_t)v;
  }
  for (int x = tid; x < n; x += nth) { scA[x] = 0; scB[x] = 0; sfirst[x] = 0x7fffffff; }
  if (__syncthreads_or(bad)) {
    if (tid == 0) atomicOr(a.status, TV_BATCH_BAD_EDGE);
  }

  // ---- node features, pads
  int flags = 0;
  for (int x = tid; x < n; x += nth) {
    const int64_t t = a.store_node_type[sn + x];
    a.node_type[n0 + x] = t;
    a.graph_id[n0 + x] = i;
    if (a.lat_row_weight) a.lat_row_weight[n0 + x] = 1.0f;
    a.node_type_pad[dbase + x] = t;
    double mu = 0.0, sd = 1.0;
    if (t < 0 || t >= a.V) {
      flags |= TV_BATCH_BAD_NODE_TYPE;
    } else {
      mu = a.lat_mean[t];
      sd = a.lat_std[t];
      if (mu != mu || sd != sd) { flags |= TV_BATCH_MISSING_LATENCY_RANGE; mu = 0.0; sd = 1.0; }
    }
    const float la = normalize_latency(a.store_avg[sn + x], mu, sd);
    const float lmin = normalize_latency(a.store_min[sn + x], mu, sd);
    const float lmax = normalize_latency(a.store_max[sn + x], mu, sd);
    // latency = [avg, min, max]  (dataset.py:80-87)
    a.latency[3 * (n0 + x) + 0] = la;
    a.latency[3 * (n0 + x) + 1] = lmin;
    a.latency[3 * (n0 + x) + 2] = lmax;
    a.latency_pad[dbase + x] = la;
  }
  if (flags) atomicOr(a.status, flags);
  if (tid == 0) a.node_count[i] = n;

  // ---- COO (edge-id order) + observed adjacency; degree counts; is the list in BFS order?
  const int64_t T = M * (M - 1) / 2;
  int unordered = 0;
  for (int e = tid; e < ne; e += nth) {
    const int64_t u = su[e], v = sv[e];
    a.enc_src[E0 + e] = n0 + u;           a.enc_dst[E0 + e] = n0 + v;
    a.enc_src[E0 + ne + e] = n0 + v;      a.enc_dst[E0 + ne + e] = n0 + u;
    a.dec_src[D0 + e] = dbase + u;        a.dec_dst[D0 + e] = dbase + v;
    a.dec_src[D0 + ne + e] = dbase + v;   a.dec_dst[D0 + ne + e] = dbase + u;
    // strict upper triangle, row-major: row u starts at u*(2M-u-1)/2  (tensor_utils.py:60-67)
    a.adj[i * T + u * (2 * M - u - 1) / 2 + (v - u - 1)] = 1.0f;
    atomicAdd(&scA[v], 1);
    atomicAdd(&scB[u], 1);
    atomicMin(&sfirst[u], e);
    if (e > 0 && (sv[e - 1] >= v || su[e - 1] > u)) unordered = 1;
  }
  for (int x = tid; x < n; x += nth) {
    a.enc_src[E0 + 2 * ne + x] = n0 + x;
    a.enc_dst[E0 + 2 * ne + x] = n0 + x;
  }
  for (int64_t x = tid; x < M; x += nth) {
    a.dec_src[D0 + 2 * ne + x] = dbase + x;
    a.dec_dst[D0 + 2 * ne + x] = dbase + x;
  }
  unordered = __syncthreads_or(unordered);

  // ---- exclusive scan of the in-degrees (wave 0, chunks of 64)
  if (tid < 64) {
    int carry = 0;
    for (int base = 0; base < n; base += 64) {
      const int x = base + lane;
      const int deg = x < n ? scA[x] + scB[x] + 1 : 0;
      int incl = deg;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      if (x < n) sptr[x] = carry + incl - deg;
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) sptr[n] = carry;  // == 3n-2
  }
  __syncthreads();

  // ---- fill, one edge per lane: both CSC entries of the edge and their reverse links
  for (int e = tid; e < ne; e += nth) {
    const int u = su[e], v = sv[e];
    int rA = 0, rB = e - sfirst[u];
    if (unordered) {
      rB = 0;
      for (int f = 0; f < e; ++f) { rA += sv[f] == v; rB += su[f] == u; }
    }
    const int pA = sptr[v] + rA, pB = sptr[u] + scA[u] + rB;
    snbr[pA] = u; seid[pA] = e;      srev[pA] = pB;
    snbr[pB] = v; seid[pB] = ne + e; srev[pB] = pA;
  }
  for (int x = tid; x < n; x += nth) {
    const int p = sptr[x + 1] - 1;   // the self loop closes the row
    snbr[p] = x; seid[p] = 2 * ne + x; srev[p] = p;
    a.enc_indptr[n0 + x] = E0 + sptr[x];
    a.dec_indptr[dbase + x] = D0 + sptr[x];
  }
  for (int64_t x = n + tid; x < M; x += nth) {
    const int64_t p = D0 + (3 * (int64_t)n - 2) + (x - n);
    a.dec_indptr[dbase + x] = p;
    a.dec_indices[p] = dbase + x;
    a.dec_eids[p] = D0 + 2 * ne + x;
    a.dec_rev[p] = p;
  }
  if (last && tid == 0) {
    a.enc_indptr[a.N] = E0 + 3 * (int64_t)n - 2;
    a.dec_indptr[a.B * M] = D0 + 2 * (int64_t)ne + M;
  }
  __syncthreads();

  // ---- CSC out (encoder and decoder share the local structure), one entry per lane
  const int np = 3 * n - 2;
  for (int p = tid; p < np; p += nth) {
    const int s = snbr[p], id = seid[p], q = srev[p];
    a.enc_indices[E0 + p] = n0 + s;
    a.enc_eids[E0 + p] = E0 + id;
    a.enc_rev[E0 + p] = E0 + q;
    a.dec_indices[D0 + p] = dbase + s;
    a.dec_eids[D0 + p] = D0 + id;
    a.dec_rev[D0 + p] = D0 + q;
  }
}

}  // namespace

extern "C" int tv_batch_build(
    int64_t batch, int64_t max_node_count, int64_t total_nodes, int64_t num_operations,
    const int64_t *node_offsets, const int64_t *store_node_start, const int64_t *store_edge_start,
    const int64_t *store_u, const int64_t *store_v, const int64_t *store_node_type,
    const float *store_avg_latency, const float *store_min_latency, const float *store_max_latency,
    const double *latency_mean, const double *latency_std, int64_t *enc_src, int64_t *enc_dst,
    int64_t *enc_indptr, int64_t *enc_indices, int64_t *enc_eids, int64_t *enc_rev,
    int64_t *dec_src, int64_t *dec_dst, int64_t *dec_indptr, int64_t *dec_indices,
    int64_t *dec_eids, int64_t *dec_rev, float *adj, int64_t *node_type_pad, float *latency_pad,
    int64_t *node_count, int64_t *node_type, float *latency, int64_t *graph_id,
    float *lat_row_weight, int32_t *status, void *stream) {
  TV_REQUIRE(batch >= 0 && total_nodes >= batch, "tv_batch_build: bad batch/total_nodes");
  TV_REQUIRE(max_node_count >= 2 && max_node_count <= 1536,
             "tv_batch_build: max_node_count must be in [2, 1536]: got %lld",
             (long long)max_node_count);
  TV_REQUIRE(num_operations >= 1, "tv_batch_build: num_operations must be >= 1");
  TV_REQUIRE(node_offsets && store_node_start && store_edge_start && store_u && store_v &&
                 store_node_type && store_avg_latency && store_min_latency && store_max_latency &&
                 latency_mean && latency_std,
             "tv_batch_build: null input");
  TV_REQUIRE(enc_src && enc_dst && enc_indptr && enc_indices && enc_eids && enc_rev && dec_src &&
                 dec_dst && dec_indptr && dec_indices && dec_eids && dec_rev && adj &&
                 node_type_pad && latency_pad && node_count && node_type && latency && graph_id &&
                 status,
             "tv_batch_build: null output");
  hipStream_t s = (hipStream_t)stream;
  const int64_t M = max_node_count;
  const int64_t T = M * (M - 1) / 2;
  if (batch == 0) {
    // empty batch: only the two terminators exist
    TV_HIP(hipMemsetAsync(enc_indptr, 0, sizeof(int64_t), s));
    TV_HIP(hipMemsetAsync(dec_indptr, 0, sizeof(int64_t), s));
    return TV_OK;
  }
  TV_HIP(hipMemsetAsync(adj, 0, sizeof(float) * batch * T, s));
  TV_HIP(hipMemsetAsync(node_type_pad, 0, sizeof(int64_t) * batch * M, s));
  TV_HIP(hipMemsetAsync(latency_pad, 0, sizeof(float) * batch * M, s));
  BatchArgs a;
  a.B = batch; a.M = M; a.N = total_nodes; a.V = num_operations;
  a.node_offsets = node_offsets; a.store_node_start = store_node_start;
  a.store_edge_start = store_edge_start; a.store_u = store_u; a.store_v = store_v;
  a.store_node_type = store_node_type; a.store_avg = store_avg_latency;
  a.store_min = store_min_latency; a.store_max = store_max_latency;
  a.lat_mean = latency_mean; a.lat_std = latency_std;
  a.enc_src = enc_src; a.enc_dst = enc_dst; a.enc_indptr = enc_indptr;
  a.enc_indices = enc_indices; a.enc_eids = enc_eids; a.enc_rev = enc_rev;
  a.dec_src = dec_src; a.dec_dst = dec_dst; a.dec_indptr = dec_indptr;
  a.dec_indices = dec_indices; a.dec_eids = dec_eids; a.dec_rev = dec_rev;
  a.adj = adj; a.node_type_pad = node_type_pad; a.latency_pad = latency_pad;
  a.node_count = node_count; a.node_type = node_type; a.latency = latency;
  a.graph_id = graph_id; a.lat_row_weight = lat_row_weight; a.status = status;
  const size_t lds_bytes = sizeof(int32_t) * (size_t)(BB_LDS_INTS_PER_NODE * M + 1);
  if (lds_bytes > 64 * 1024) {
    static bool raised[64] = {};         // per device: the attribute belongs to the device's code object
    int device = 0;
    TV_HIP(hipGetDevice(&device));
    if (!raised[device & 63]) {
      TV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&batch_build_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised[device & 63] = true;
    }
  }
  // long traces get four waves per trace, short ones one
  const unsigned threads = M > 128 ? 256 : 64;
  hipLaunchKernelGGL(batch_build_kernel, dim3((unsigned)batch), dim3(threads), lds_bytes, s, a);
  TV_LAUNCH_CHECK("tv_batch_build");
  return TV_OK;
}
