/*
 * tracevae_hip.h - C-ABI of libtracevae_hip.so: the MI355X (gfx950) kernels behind the
 * TraceVAE training / scoring path.
 *
 * The reference (NetManAIOps/TraceVAE) is pure Python and has no FFI of its own: its seam
 * is the Python model API, and the arithmetic is done by DGL / PyTorch operators called
 * from Python.  Every entry point below therefore cites the reference *call site* (file:line
 * under /root/reference) whose DGL / torch / tensorkit operator it replaces.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless the
 *     comment says "host";
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - the caller owns every buffer: kernels never allocate, workspaces are caller-provided
 *     and their sizes come from the matching *_workspace_bytes call;
 *   - return value: 0 = ok, <0 = error (TV_ERR_*); tv_last_error() gives the text for the
 *     calling thread.  Shapes are validated on the host before any launch;
 *   - every reduction is a fixed-order segmented / two-stage reduction: no float atomics,
 *     results are bitwise reproducible run to run;
 *   - index arrays are int64 (DGL's default idtype), features are float32.
 */
#ifndef TRACEVAE_HIP_H
#define TRACEVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TV_OK 0
#define TV_ERR_INVALID (-1) /* bad argument / shape */
#define TV_ERR_LAUNCH (-2)  /* HIP reported a launch error */
#define TV_ERR_WORKSPACE (-3) /* workspace too small */

/* limits of tv_train_loss_* */
#define TV_LOSS_MAX_VECTORS 8
#define TV_LOSS_MAX_LATENTS 2

/* limits of tv_realnvp_logprob_* */
#define TV_FLOW_MAX_LEVELS 8
#define TV_FLOW_MAX_DIM 20
/* partial operand maxima kept by the caller of tv_gemm_f32_ex */
#define TV_GEMM_MAX_SLOTS 256

/* status bits written by tv_batch_build into *status (device int32) */
#define TV_BATCH_TOO_MANY_NODES 1 /* a trace has more than max_node_count nodes           */
#define TV_BATCH_BAD_EDGE 2       /* an edge violates 0 <= u < v < n                        */
#define TV_BATCH_MISSING_LATENCY_RANGE 4 /* operation without (mean,std): reference KeyError */
#define TV_BATCH_BAD_NODE_TYPE 8  /* node_type outside [0, num_operations)                  */
/* status bit written by tv_sampled_graph_count */
#define TV_SAMPLED_BAD_ADJ 16     /* a sampled adjacency entry is neither 0 nor 1           */

const char *tv_last_error(void);
int tv_version(void);

/* number of partial-result slots the two-stage column reductions use for `rows` rows */
int64_t tv_stat_blocks(int64_t rows);

/* ------------------------------------------------------------------------------------
 * (a) batching: collated trace graphs -> device-side COO + destination-sorted CSC
 *
 * Replaces trace_graph_to_dgl (tracegnn/models/trace_vae/dataset.py:20-89: dgl.graph +
 * add_self_loop + ndata + latency normalisation), TraceGraphBatch.build_dgl
 * (types.py:42-62: dgl.batch(...).to(device)), dense_triu_adj + dense_to_triu
 * (tensor_utils.py:79-89, 64-67; stacked at model/trace_vae.py:122-132),
 * pad_node_feature (tensor_utils.py:92-114; model/trace_vae.py:134-140) and the padded
 * decoder graph rebuilt on every p() call (model/struct_vae.py:335-362).
 *
 * Input: a trace store in HBM (concatenated per-trace arrays; u, v are trace-local node
 * ids with u < v) and, for the B traces of this batch, where each one starts in the store.
 * Trace i has n_i = node_offsets[i+1]-node_offsets[i] nodes and n_i-1 edges.
 *
 * Output (N = node_offsets[B], E_enc = 3N-2B, E_dec = B*M + 2(N-B)):
 *   enc_src/enc_dst [E_enc]  COO in DGL edge-id order: per trace [u->v ; v->u ; self x n]
 *   enc_indptr [N+1], enc_indices [E_enc] (=src), enc_eids [E_enc]: stable sort by dst
 *   enc_rev [E_enc]: CSC position of the opposite edge (used by the backward pass)
 *   dec_*: the same for the decoder graph, every trace padded to M nodes:
 *          per trace [u->v ; v->u ; self x M], node ids i*M + local
 *   adj [B, M(M-1)/2] f32, node_type_pad [B,M] i64, latency_pad [B,M] f32 (avg latency),
 *   node_count [B] i64, node_type [N] i64, latency [N,3] f32 (avg,min,max normalised as
 *   float32((double(x) - mean[op]) / (std[op] + 1e-5))), graph_id [N] i64.
 *
 * lat_row_weight != NULL additionally emits the COMPACT latency-decoder graph (the padded
 * slots of one trace carry identical rows through model/latency_vae.py:320-343, so one
 * representative per trace stands for its M - n_i slots): rows [real nodes (N) | one per
 * trace (B)], i.e. the encoder graph plus B self-loop nodes.  The caller then sizes
 * enc_src/enc_dst/enc_indices/enc_eids/enc_rev with E_enc + B entries, enc_indptr with
 * N + B + 1, node_type and graph_id with N + B, and lat_row_weight [N + B] receives the
 * BatchNorm weight of each row (1 for a node, M - n_i for a representative).
 * ---------------------------------------------------------------------------------- */
int tv_batch_build(int64_t batch, int64_t max_node_count, int64_t total_nodes,
                   int64_t num_operations,
                   const int64_t *node_offsets,   /* [B+1] batch-local, node_offsets[0]=0 */
                   const int64_t *store_node_start, /* [B] */
                   const int64_t *store_edge_start, /* [B] */
                   const int64_t *store_u, const int64_t *store_v,
                   const int64_t *store_node_type, const float *store_avg_latency,
                   const float *store_min_latency, const float *store_max_latency,
                   const double *latency_mean, const double *latency_std, /* [V], NaN = missing */
                   int64_t *enc_src, int64_t *enc_dst, int64_t *enc_indptr,
                   int64_t *enc_indices, int64_t *enc_eids, int64_t *enc_rev,
                   int64_t *dec_src, int64_t *dec_dst, int64_t *dec_indptr,
                   int64_t *dec_indices, int64_t *dec_eids, int64_t *dec_rev, float *adj,
                   int64_t *node_type_pad, float *latency_pad, int64_t *node_count,
                   int64_t *node_type, float *latency, int64_t *graph_id,
                   float *lat_row_weight, /* nullable, see above */
                   int32_t *status, /* device int32, OR-ed TV_BATCH_* bits; caller zeroes it */
                   void *stream);

/* ------------------------------------------------------------------------------------
 * decoder graphs of traces SAMPLED from the prior (section 8f row 4)
 *
 * Replaces make_graph in TraceStructVAE.p (model/struct_vae.py:296-333: triu_to_dense +
 * T.where + dgl.graph + dgl.add_reverse_edges + dgl.add_self_loop + the node_idx formula,
 * then dgl.batch over the samples).  adj [G, M(M-1)/2] f32 holds 0/1 draws of the strict
 * upper triangle, row-major (tensor_utils.py:60-67); any DAG, not only trees.
 *
 * Two calls, because the edge total is data dependent:
 *   tv_sampled_graph_count -> indptr [G*M+1] (final values) and graph_edge_offsets [G+1];
 *       the caller reads E = graph_edge_offsets[G] back and sizes the arrays below with it;
 *   tv_sampled_graph_fill  -> src/dst [E] COO in DGL edge-id order, per graph
 *       [u->v in row-major order ; the same reversed ; self x M], node ids g*M + local;
 *       indices/eids [E] = stable sort by destination; rev [E] = CSC position of the
 *       opposite edge; node_idx [G*M] = max(max_u(adj[u,v] * cumsum_v(adj[u,:])) - 1, 0).
 * Both take the same workspace (tv_sampled_graph_workspace_bytes); fill must follow count on
 * the same adjacency (a graph whose edge total no longer matches is left unwritten).
 * ---------------------------------------------------------------------------------- */
int64_t tv_sampled_graph_workspace_bytes(int64_t graphs, int64_t max_node_count);
int tv_sampled_graph_count(int64_t graphs, int64_t max_node_count, const float *adj,
                           int64_t *indptr, int64_t *graph_edge_offsets, void *ws,
                           int64_t ws_bytes,
                           int32_t *status, /* device int32, OR-ed TV_SAMPLED_*; caller zeroes */
                           void *stream);
int tv_sampled_graph_fill(int64_t graphs, int64_t max_node_count, int64_t num_edges,
                          const float *adj, const int64_t *graph_edge_offsets,
                          const int64_t *indptr, int64_t *src, int64_t *dst, int64_t *indices,
                          int64_t *eids, int64_t *rev, int64_t *node_idx, void *ws,
                          int64_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * dense layers: C[M,N] = row_scale[m] * (op(A)[M,K] . op(B)[K,N]) + bias[n]
 * fp32 in / fp32 out, fp32-accurate on the 16-bit matrix cores, in one of two forms chosen by
 * shape (tv_gemm_f32_last_form() tells which one the calling thread's last product took):
 *   form 1, 2 - the layer products and weight gradients of many rows (128 x 256 tiles): each
 *     operand is scaled by a power of two taken from its largest magnitude (a pass over the
 *     operand), split into TWO fp16 pieces x0 + 2^-11 x1 (11 + 11 bits), and the three partial
 *     products a0 b0 | a0 b1 + a1 b0 are accumulated in fp32 on v_mfma_f32_32x32x16_f16;
 *   form 0 - every other shape: three bf16 pieces per element, six partial products on
 *     v_mfma_f32_32x32x16_bf16, no scaling.
 * Either way the error against float64 is below that of an fp32 fma chain
 * (tests/test_gpu_ops.py::test_gemm_error_against_float64...).  An infinite operand element
 * gives NaN where an fp32 chain gives inf.
 * row-major storage; transX = 0: stored as written, 1: stored transposed.  lda, ldb < 2^22.
 *
 * Replaces nn.Linear inside dgl.nn.GATConv.fc (model/gnn_layers.py:134-143), the matmul
 * of dgl.nn.GraphConv (gnn_layers.py:173-183), tk.layers.Linear / Dense
 * (model/struct_vae.py:139-180, model/latency_vae.py:181-203) and their autograd
 * backward (dX = dY.W, dW = dY^T.X use the same kernel with other trans flags; a
 * reduction over many rows is split over K with a fixed-order second-stage sum).
 * bias / row_scale may be NULL.  accumulate != 0 adds into C.
 * Workspace: tv_gemm_f32_workspace_bytes(m, n, k) bytes of 16-byte aligned device memory (0 for
 * many shapes), used for the split-K slabs of weight-gradient shapes, for the K-split of the
 * output tiles of a partial last round of workgroups (both summed in a fixed order by a second
 * kernel), and for the bf16 pieces of a small B that is split once per call (m >= 2048,
 * n * k <= 2^25: 6 bytes per padded element, at most ~200 MB).
 * A row-major A whose k is not a multiple of 4 is still read 16 bytes at a time when lda is a
 * multiple of 4, lda >= k rounded up to 4 and a is 16-byte aligned (the floats between k and
 * the rounded k must be readable; their values do not matter).
 * ---------------------------------------------------------------------------------- */
int64_t tv_gemm_f32_workspace_bytes(int64_t m, int64_t n, int64_t k);
int tv_gemm_f32_last_form(void);
/* B split ahead of its products, several operands per launch (no counterpart in the reference:
 * its GEMMs are torch's).  tv_gemm_b_form: the form of B's plane image in a plain product
 * C[m,n] = A[m,k] . op(B) with a row-major, 16-byte readable A - -1: B is split inside the
 * kernel (not preparable), 0: three bf16 planes, 1: two fp16 planes behind the operand's
 * power-of-two scale.  tv_gemm_prepared_b_bytes: size of that image.  tv_gemm_prepare_b_batch:
 * writes the images of `count` (<= 32) operands in one launch; the arrays are HOST arrays of
 * `count` entries (m / n / k / trans_b / b / ldb of each product, b_max = the operand's
 * TV_GEMM_MAX_SLOTS partial maxima on the device - needed for form 1 -, planes = where its image
 * goes, 16-byte aligned).  The images are then passed as `b` with b_max_state 3 / 4. */
int tv_gemm_b_form(int64_t m, int64_t n, int64_t k);
int64_t tv_gemm_prepared_b_bytes(int64_t m, int64_t n, int64_t k);
int tv_gemm_prepare_b_batch(int64_t count, const int64_t *m, const int64_t *n, const int64_t *k,
                            const int *trans_b, const float *const *b, const int64_t *ldb,
                            const float *const *b_max, void *const *planes, void *stream);
/* Measurement aid, no counterpart in the reference.  Between _begin and _end the MAIN kernel of
 * every product (not the operand scans, the split of B or the reductions around it) is bracketed
 * by a pair of HIP events on the stream it is launched on.  _end waits for them, writes
 * [milliseconds, form, m, n, k] as 5 doubles per product in launch order (up to `capacity`
 * records; `records` is HOST memory) and returns the number of products seen.  bench.py prices
 * the dominant kernel with it; rocprofv3's per-kernel averages are the cross-check. */
int tv_gemm_profile_begin(void);
int64_t tv_gemm_profile_end(double *records, int64_t capacity);
int tv_gemm_f32(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float *a,
                int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc,
                const float *bias, const float *row_scale, int accumulate, void *workspace,
                int64_t workspace_bytes, void *stream);
/* The same product with BatchNorm + LeakyReLU applied to one operand on its way into LDS, so the
 * activation between two layers (model/gnn_layers.py:152-155 feeding the next GATConv.fc,
 * gnn_layers.py:134-143) is never written to memory:
 *   act_operand 1: A (needs trans_a = 0), A[m,k] -> lrelu(A[m,k] * act_scale[k] + act_shift[k])
 *                  - the forward product  ft = act(x) . W^T
 *   act_operand 2: B (needs trans_b = 0), B[k,n] -> lrelu(B[k,n] * act_scale[n] + act_shift[n])
 *                  - the weight gradient  dW = dft^T . act(x)
 *   act_operand 0: plain tv_gemm_f32.
 * act_scale / act_shift: 16-byte aligned device vectors (tv_bn_finalize_coeffs writes them).
 * Same workspace rule as tv_gemm_f32. */
int tv_gemm_f32_act(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float *a,
                    int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc,
                    const float *bias, const float *row_scale, int accumulate, int act_operand,
                    const float *act_scale, const float *act_shift, float act_slope,
                    void *workspace, int64_t workspace_bytes, void *stream);

/* tv_gemm_f32_act with the operand maxima of the fp16 forms kept by the CALLER, so that a tensor
 * that enters several products is scanned once (a layer's input x enters its forward product
 * and its weight gradient; the output gradient enters the input gradient and the weight
 * gradient; the weights enter forward product and input gradient).  a_max / b_max: 256 floats of
 * device memory each (TV_GEMM_MAX_SLOTS; partial maxima of |operand| behind its prologue).
 * state 0: no buffer (the library scans into its workspace); 1: scan the operand into the buffer
 * if this product needs it; 2: the buffer holds the maxima of exactly this operand view (same
 * rows, columns and prologue coefficients) - the library does not look at the operand again.
 * Whether a state-1 buffer was filled: tv_gemm_f32_last_form() != 0 after the call.
 * For B two more states: 3 / 4 = `b` is not the operand but its plane image, written ahead of
 * the product by tv_gemm_prepare_b_batch in the bf16 (3) / fp16 (4) form, and b_max (state 4)
 * holds the maxima the image was scaled by; ldb is then only checked against n / k.  The form
 * must be the one this product takes (tv_gemm_b_form), else the call fails.
 * No counterpart in the reference (its GEMMs are torch's); the shapes are those of
 * tv_gemm_f32_act. */
int tv_gemm_f32_ex(int trans_a, int trans_b, int64_t m, int64_t n, int64_t k, const float *a,
                   int64_t lda, const float *b, int64_t ldb, float *c, int64_t ldc,
                   const float *bias, const float *row_scale, int accumulate, int act_operand,
                   const float *act_scale, const float *act_shift, float act_slope, float *a_max,
                   int a_max_state, float *b_max, int b_max_state, void *workspace,
                   int64_t workspace_bytes, void *stream);

/* Linear layers with one small side, min(in_features, out_features) <= 16 - the z / z2 mean and
 * log-std heads (1000 -> 10) and the first layers fed by z (10 -> 500 / 250):
 * model/struct_vae.py:139-180, model/latency_vae.py:181-203.  y = x W^T + bias with W
 * [out_features, in_features] as torch / tensorkit Linear stores it.  At 512 rows these products
 * are far too small for the 128x128 MFMA tiles.  Backward: dx (nullable) and
 * dwb = [dW (out*in) | dbias (out)] in one buffer; fixed-order two-stage reduction over rows. */
int tv_skinny_linear_fwd(int64_t rows, int64_t in_features, int64_t out_features, const float *x,
                         const float *w, const float *bias /* nullable */, float *out,
                         void *stream);
int64_t tv_skinny_linear_bwd_workspace_bytes(int64_t rows, int64_t in_features,
                                             int64_t out_features);
int tv_skinny_linear_bwd(int64_t rows, int64_t in_features, int64_t out_features, const float *x,
                         const float *w, const float *dout, float *dx /* nullable */, float *dwb,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* skinny right-hand side (N <= 8): out[r,j] = row_scale[r] * sum_k x[r,k] w[k,j]  (w is [K,N])
 * Replaces the 1000->2 GraphConv matmul of the latency decoder (model/latency_vae.py:221-229,
 * DGL multiplies by W before aggregating when in > out). */
int tv_rowdot_fwd(int64_t rows, int64_t k, int64_t n, const float *x, const float *w,
                  const float *row_scale, float *out, void *stream);
int64_t tv_rowdot_bwd_workspace_bytes(int64_t rows, int64_t k, int64_t n);
int tv_rowdot_bwd(int64_t rows, int64_t k, int64_t n, const float *x, const float *w,
                  const float *row_scale, const float *dout, float *dx /* nullable */,
                  float *dw /* [K,N] */, void *workspace, int64_t workspace_bytes, void *stream);

/* column sums: out[c] = sum_r x[r,c]  (bias gradients).  Two-stage, fixed order. */
int64_t tv_colsum_workspace_bytes(int64_t rows, int64_t cols);
int tv_colsum(int64_t rows, int64_t cols, const float *x, int64_t ldx, float *out,
              void *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * (b) GAT message passing over the destination-sorted CSC.
 * The graph may be replicated `n_rep` times block-diagonally (test-time K samples):
 * row r = rep * rows_per_rep + v uses the CSC of node v, neighbours offset by rep*rows_per_rep.
 *
 * tv_gat_scores: el[r,h] = sum_f ft[r,h,f] attn_l[h,f], er likewise
 *   (the (ft*attn).sum(-1) of dgl.nn.GATConv; call site model/gnn_layers.py:143).
 * tv_gat_aggregate_fwd: e = LeakyReLU_0.2(el[src]+er[dst]); a = softmax over the in-edges
 *   of dst; out[dst] = sum a * ft[src] + bias   (apply_edges + edge_softmax + update_all of
 *   GATConv, gnn_layers.py:143), heads concatenated as [rows, H*F] (gnn_layers.py:144-148).
 *   alpha [n_rep * E, H] is kept in CSC order for the backward pass (NULL to skip).
 *   Optional fused epilogues:
 *     stats_partial != NULL : per-block column sum / sum of squares of `out`
 *                             [tv_stat_blocks(rows), 2, H*F] for the BatchNorm that follows
 *                             (gnn_layers.py:152-153, training mode); row v counts
 *                             stats_row_weight[v] times when that is not NULL;
 *     ep_scale/ep_shift != NULL : y = LeakyReLU_slope(out*ep_scale + ep_shift) written to
 *                             `out` instead (eval-mode BatchNorm + activation, :152-155).
 * tv_gat_aggregate_bwd: gradients w.r.t. ft, attn_l, attn_r, bias given d(out).
 * tv_gat_fc_score_weight: weight_ext [H*F + pad4(2H), in] = [W ; A_l W ; A_r W ; 0], row H*F + h =
 *   sum_f attn_l[h,f] W[h*F+f, :] (then attn_r): el = (x W^T) . attn_l = x . (W^T attn_l), so the
 *   fc product of GATConv (gnn_layers.py:143, `self.fc(h)`) with weight_ext gives ft, el and er
 *   in one pass [rows, H*F | H | H] and the rows of ft are not read again to form the scores.
 * tv_gat_fc_score_weight_bwd: its backward.  With `score_grad_columns` = 1 tv_gat_aggregate_bwd
 *   writes [dft | del | der | 0] (dft WITHOUT the el / er terms, dattn_l / dattn_r untouched, may be
 *   NULL): the two GEMMs of the fc layer with weight_ext then carry the score gradients to x and to
 *   dweight_ext [H*F + pad4(2H), in] = [dft | del | der]^T x, and this call folds its derived rows
 *   back: dW += attn_l (x) dWx[H*F+h] + attn_r (x) dWx[H*F+H+h] (in place, rows 0..H*F-1),
 *   dattn_l[h,f] = <W[h*F+f], dWx[H*F+h]>.  The backward pass then does not read ft's rows again.
 * out_max (forward, with the eval epilogue only) / dft_max (backward, with score_grad_columns
 *   only), nullable: TV_GEMM_MAX_SLOTS floats that receive partial maxima of |out| / of the
 *   whole [dft | del | der | 0] rows, for the product that reads those rows next
 *   (tv_gemm_f32_ex, state 2); folded in with integer atomic max (order-free, deterministic).
 * Row strides (in floats) let the aggregation read that layout in place: `ft_stride` of ft
 *   (0 = H*F; a multiple of 4), `score_stride` of el / er (0 = H), `dft_stride` of dft (0 = H*F).
 * ---------------------------------------------------------------------------------- */
int tv_gat_fc_score_weight(int64_t heads, int64_t feats, int64_t in_feats, const float *weight,
                           const float *attn_l, const float *attn_r, float *weight_ext,
                           float *weight_ext_max, /* nullable: TV_GEMM_MAX_SLOTS partial maxima of
                                                     |weight_ext|, for tv_gemm_f32_ex state 2 */
                           void *stream);
/* the same for up to 16 layers in one launch (HOST arrays of `count` entries; every layer's
 * ceil(in_feats / 16) * heads must be <= 256: its maxima go to one slot per workgroup) */
int tv_gat_fc_score_weight_batch(int64_t count, const int64_t *heads, const int64_t *feats,
                                 const int64_t *in_feats, const float *const *weight,
                                 const float *const *attn_l, const float *const *attn_r,
                                 float *const *weight_ext, float *const *weight_ext_max,
                                 void *stream);
int tv_gat_fc_score_weight_bwd(int64_t heads, int64_t feats, int64_t in_feats,
                               const float *weight, const float *attn_l, const float *attn_r,
                               float *dweight_ext, float *dattn_l, float *dattn_r, void *stream);
int tv_gat_scores(int64_t rows, int64_t heads, int64_t feats, const float *ft,
                  const float *attn_l, const float *attn_r, float *el, float *er, void *stream);
/* (tv_gat_aggregate_fwd: large inference launches - the [ft | el | er] layout, no statistics,
 * at least 16384 rows of at most 1024 floats - run on a windowed form of the kernel that stages
 * 32 consecutive rows per workgroup in LDS and serves the gathers inside the window from there;
 * bit-equal to the per-row form.  TV_GAT_LDS_MIN_WINDOWS in the environment moves the threshold,
 * for A/B measurements.) */
int tv_gat_aggregate_fwd(int64_t rows_per_rep, int64_t n_rep, int64_t heads, int64_t feats,
                         const int64_t *indptr, const int64_t *indices, const float *ft,
                         const float *el, const float *er, const float *bias, float *out,
                         float *alpha, float *stats_partial, const float *stats_row_weight,
                         const float *ep_scale, const float *ep_shift, float ep_slope,
                         int64_t ft_stride, int64_t score_stride, float *out_max, void *stream);
int64_t tv_gat_aggregate_bwd_workspace_bytes(int64_t rows_per_rep, int64_t n_rep,
                                             int64_t heads, int64_t feats, int64_t num_edges);
int tv_gat_aggregate_bwd(int64_t rows_per_rep, int64_t n_rep, int64_t heads, int64_t feats,
                         int64_t num_edges, const int64_t *indptr, const int64_t *indices,
                         const int64_t *rev, const float *ft, const float *el, const float *er,
                         const float *attn_l, const float *attn_r, const float *alpha,
                         const float *dout, float *dft, float *dattn_l, float *dattn_r,
                         float *dbias, int64_t ft_stride, int64_t score_stride,
                         int64_t dft_stride, int score_grad_columns, float *dft_max,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * One GATConvAgg layer of a TRAINING step as one call each way (model/gnn_layers.py:112-157:
 * GATConv, head concat; the BatchNorm + LeakyReLU that FOLLOWS the previous layer rides in this
 * layer's products).  The two calls issue exactly the launches of the operator-level entry
 * points above, in the same order with the same arguments (bit-equal results); they exist for
 * hosts whose cost per crossing of this boundary is of the order of a kernel's run time (a
 * Python host at the reference's batch size of 64 traces).
 * Forward: [tv_bn_finalize_coeffs of the input's statistics] -> tv_gat_fc_score_weight ->
 *   tv_gemm_f32_ex (A prologue when the input is deferred) -> tv_gat_aggregate_fwd.
 *   in_coeffs == NULL: x is a plain input; else [4, pad4(in_feats)] floats that receive mean,
 *   rstd, scale, shift of the input's BatchNorm (in_stats [in_nblk, 2, in_feats], in_count rows).
 *   weight_ext [H*F + pad4(2H), in_feats], ft [rows, H*F + pad4(2H)], out [rows, H*F],
 *   alpha [E, H], out_stats [tv_stat_blocks(rows), 2, H*F] (NULL: none) are written;
 *   x_max / w_max: TV_GEMM_MAX_SLOTS floats each, as in tv_gemm_f32_ex (x_max_state for a plain
 *   input: 1 or 2; behind a BatchNorm the finalize writes the bound).
 * Backward: tv_gat_aggregate_bwd (score-gradient columns) -> weight gradient product ->
 *   tv_gat_fc_score_weight_bwd -> input gradient product -> [tv_bn_lrelu_bwd].
 *   dweight_ext [H*F + pad4(2H), in_feats]: rows 0..H*F-1 are dW on return.  dx [rows, in_feats];
 *   dgamma / dbeta [in_feats] with a deferred input.  need_dx / need_dw: skip the products whose
 *   results nobody reads.  Workspace: tv_gat_layer_workspace_bytes(...) bytes, 256-byte aligned.
 * Prepared weights: the weights of a stack change once per step, so a host may build every
 * layer's extended weight and its plane images - forward orientation for the fc product,
 * transposed for the input gradient - with ONE tv_gat_fc_score_weight_batch and ONE
 * tv_gemm_prepare_b_batch call per stack instead of three launches per layer.  Then
 * `weight_ext` is that image and w_prepared (forward) / w_max_state (backward) is 3 or 4, the
 * form tv_gemm_b_form reports for the product (tv_gemm_f32_ex states of B); w_max holds the
 * maxima tv_gat_fc_score_weight_batch left.  0 / 1 / 2: the layer builds them itself, as before.
 * ---------------------------------------------------------------------------------- */
int64_t tv_gat_layer_workspace_bytes(int64_t rows, int64_t in_feats, int64_t heads,
                                     int64_t feats, int64_t num_edges);
int tv_gat_layer_train_fwd(int64_t rows, int64_t in_feats, int64_t heads, int64_t feats,
                           const int64_t *indptr, const int64_t *indices, const float *x,
                           int64_t x_stride, const float *in_stats, int64_t in_nblk,
                           int64_t in_count, float eps, float momentum, float slope,
                           const float *in_gamma, const float *in_beta, float *in_running_mean,
                           float *in_running_var, float *in_coeffs, const float *weight,
                           const float *attn_l, const float *attn_r, const float *bias,
                           const float *stats_row_weight, float *weight_ext, float *ft,
                           float *out, float *alpha, float *out_stats, float *x_max,
                           int x_max_state, float *w_max, int w_prepared, void *workspace,
                           int64_t workspace_bytes, void *stream);
int tv_gat_layer_train_bwd(int64_t rows, int64_t in_feats, int64_t heads, int64_t feats,
                           int64_t num_edges, const int64_t *indptr, const int64_t *indices,
                           const int64_t *rev, const float *x, int64_t x_stride,
                           const float *in_coeffs, const float *in_gamma, const float *in_beta,
                           float slope, const float *in_row_weight, int64_t in_count,
                           const float *weight, const float *attn_l, const float *attn_r,
                           const float *weight_ext, const float *ft, const float *alpha,
                           const float *dout, float *x_max, int x_max_state, float *w_max,
                           int w_max_state, int need_dx, int need_dw, float *dweight_ext,
                           float *dattn_l, float *dattn_r, float *dbias, float *dx,
                           float *dgamma, float *dbeta, void *workspace,
                           int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm over rows + LeakyReLU (tk.layers.BatchNorm = torch BatchNorm1d, then
 * tk.layers.LeakyReLU; model/gnn_layers.py:152-155 and every `dense` of the MLPs,
 * model/struct_vae.py:117-120).
 *   tv_colstats        : partial[nblk,2,C] = per-block column sum / sum of squares
 *                        (row r counts row_weight[r] times when row_weight != NULL)
 *   tv_bn_finalize     : mean, rstd (biased variance) from the partials, in float64;
 *                        running_mean/var updated with `momentum` (unbiased variance)
 *   tv_bn_finalize_coeffs: the same, plus scale = gamma*rstd, shift = beta - mean*scale
 *                        (the folded form tv_gemm_f32_act / tv_affine_lrelu apply)
 *   tv_bn_apply_lrelu  : y = LeakyReLU_slope((x-mean)*rstd*gamma + beta)
 *   tv_affine_lrelu    : y = LeakyReLU_slope(x*scale + shift)   (eval mode, folded stats)
 *   tv_bn_eval_coeffs  : scale = gamma/sqrt(var+eps), shift = beta - mean*scale
 *   tv_bn_lrelu_bwd    : dx, dgamma, dbeta given dy (training mode: batch statistics);
 *                        count = rows, or the sum of row_weight when rows are weighted
 *   tv_affine_lrelu_bwd: dx = dy * lrelu'(x*scale+shift) * scale (eval mode inside a
 *                        training step: phase 2 of train.py:204-208)
 * ----------------------------------------------------------------------------------  * act_max (tv_bn_finalize_coeffs, nullable, cols <= 2048): TV_GEMM_MAX_SLOTS floats that receive
 * a bound on |LeakyReLU(x * scale + shift)| over the rows behind the statistics
 * (|gamma| sqrt(count) + |beta| per column), for the product that applies these coefficients to
 * its operand (tv_gemm_f32_ex, state 2); rows of weight 0 are not covered by it (the fp16 split
 * saturates instead of overflowing).
 */
int tv_colstats(int64_t rows, int64_t cols, const float *x, const float *row_weight,
                float *partial, void *stream);
int tv_bn_finalize(int64_t cols, int64_t nblk, const float *partial, int64_t count, float eps,
                   float momentum, float *mean, float *rstd, float *running_mean,
                   float *running_var, void *stream);
int tv_bn_finalize_coeffs(int64_t cols, int64_t nblk, const float *partial, int64_t count,
                          float eps, float momentum, float *mean, float *rstd,
                          float *running_mean, float *running_var, const float *gamma,
                          const float *beta, float *scale, float *shift, float *act_max,
                          void *stream);
int tv_bn_apply_lrelu(int64_t rows, int64_t cols, const float *x, const float *mean,
                      const float *rstd, const float *gamma, const float *beta, float slope,
                      float *y, void *stream);
int tv_bn_eval_coeffs(int64_t cols, const float *running_mean, const float *running_var,
                      const float *gamma, const float *beta, float eps, float *scale,
                      float *shift, void *stream);
int tv_affine_lrelu(int64_t rows, int64_t cols, const float *x, const float *scale,
                    const float *shift, float slope, float *y, void *stream);
int64_t tv_bn_lrelu_bwd_workspace_bytes(int64_t rows, int64_t cols);
int tv_bn_lrelu_bwd(int64_t rows, int64_t cols, const float *x, const float *dy,
                    const float *mean, const float *rstd, const float *gamma, const float *beta,
                    float slope, const float *row_weight, int64_t count, float *dx,
                    float *dgamma, float *dbeta, void *workspace, int64_t workspace_bytes,
                    void *stream);
int tv_affine_lrelu_bwd(int64_t rows, int64_t cols, const float *x, const float *dy,
                        const float *scale, const float *shift, float slope, float *dx,
                        void *stream);

/* ------------------------------------------------------------------------------------
 * GraphConv(norm='both') aggregation: out[v] = sum_{u in in(v)} x[u] * deg(u)^-1/2
 * (degrees include the self loop, clamp >= 1), optionally followed by * deg(v)^-1/2.
 * The graph is symmetric, so the backward pass is the same kernel with the roles of the
 * two scalings swapped.  Replaces the update_all(copy_u, sum) of dgl.nn.GraphConv
 * (model/gnn_layers.py:173-183; final decoder layers model/struct_vae.py:201-206,
 * model/latency_vae.py:221-229).
 * out_max (optional, TV_GEMM_MAX_SLOTS floats; needs cols % 4 == 0, cols >= 64, 16-byte aligned
 * rows): the partial maxima of |out| for the product that reads `out` next (tv_gemm_f32_ex,
 * state 2), as tv_gat_aggregate_fwd leaves them.
 * ---------------------------------------------------------------------------------- */
/* act_scale / act_shift (optional, cols floats each, same conditions as out_max): the rows of x
 * go through lrelu(v * act_scale[col] + act_shift[col], act_slope) as they are gathered - the
 * BatchNorm + LeakyReLU that ends the GAT layer in front of this one (gnn_layers.py:152-155) with
 * its coefficients folded (tv_bn_finalize_coeffs), so that its output is never written. */
int tv_graph_sum(int64_t rows_per_rep, int64_t n_rep, int64_t cols, const int64_t *indptr,
                 const int64_t *indices, const float *x, int scale_src, int scale_dst,
                 float *out, float *out_max, const float *act_scale, const float *act_shift,
                 float act_slope, void *stream);
/* deg^-1/2 per row of the (replicated) graph, for the GEMM row_scale epilogue */
int tv_graph_norm(int64_t rows_per_rep, int64_t n_rep, const int64_t *indptr, float *norm,
                  void *stream);

/* ------------------------------------------------------------------------------------
 * graph readout: dgl.nn.AvgPooling (model/pooling.py:45) = per-trace mean of node rows
 * ---------------------------------------------------------------------------------- */
/* (act_scale / act_shift / act_slope: optional prologue on the rows, as in tv_graph_sum; needs
 * cols % 4 == 0 and 16-byte aligned operands) */
int tv_segment_mean_fwd(int64_t batch, int64_t cols, const int64_t *node_offsets,
                        const float *x, float *out, const float *act_scale,
                        const float *act_shift, float act_slope, void *stream);
int tv_segment_mean_bwd(int64_t rows, int64_t cols, const int64_t *node_offsets,
                        const int64_t *graph_id, const float *dout, float *dx, void *stream);

/* operation embedding: torch.nn.Embedding row gather (model/operation_embedding.py:21-25)
 * written into columns [col0, col0+dim) of a [rows, ld] buffer; bwd: deterministic
 * per-operation segmented sum. */
int tv_embedding_fwd(int64_t rows, int64_t dim, int64_t num_operations, const float *weight,
                     const int64_t *ids, float *out, int64_t ld, int64_t col0, void *stream);
int tv_embedding_bwd(int64_t rows, int64_t dim, int64_t num_operations, const int64_t *ids,
                     const float *dout, int64_t ld, int64_t col0, float *dweight, void *stream);

/* latency-decoder input rows [z2 (broadcast over the M slots) || emb[node_type_pad]]
 * (model/latency_vae.py:320-343): out [K*B*M, Z2+E]; bwd: dz2[k,b,:] = sum_m d(out)[..., :Z2] */
int tv_latdec_input_fwd(int64_t n_samples, int64_t batch, int64_t max_node_count, int64_t z2_dim,
                        int64_t emb_dim, int64_t num_operations, const float *z2,
                        const float *emb_weight, const int64_t *node_type_pad, float *out,
                        void *stream);
int tv_latdec_input_bwd(int64_t n_samples, int64_t batch, int64_t max_node_count, int64_t z2_dim,
                        int64_t emb_dim, const float *dout, float *dz2, void *stream);
/* the same on compact rows (see tv_batch_build): replica k holds rows_per_rep = N + B rows,
 * row i belongs to trace row_graph[i] and has operation row_type[i];
 * out[k*R + i] = [z2[k, row_graph[i]] || emb[row_type[i]]]; bwd sums a trace's node rows
 * (node_offsets) and its representative row N + b. */
int tv_latdec_rows_fwd(int64_t n_samples, int64_t rows_per_rep, int64_t batch, int64_t z2_dim,
                       int64_t emb_dim, int64_t num_operations, const float *z2,
                       const float *emb_weight, const int64_t *row_graph, const int64_t *row_type,
                       float *out, void *stream);
int tv_latdec_rows_bwd(int64_t n_samples, int64_t rows_per_rep, int64_t batch, int64_t z2_dim,
                       int64_t emb_dim, const int64_t *node_offsets, const float *dout, float *dz2,
                       void *stream);

/* ------------------------------------------------------------------------------------
 * (c)/(d) likelihoods
 * ---------------------------------------------------------------------------------- */
/* p(adj|z): edge_logits_by_dot_product + dense_to_triu + Bernoulli(event_ndims=1) log-prob
 * (tensor_utils.py:54-67, model/struct_vae.py:281-294), never materialising [K,B,M,M]:
 * ll[k,b] = sum_{i<j} f(adj*l - softplus(l)), l = <h[k,b,i,:], h[k,b,j,:]>,
 * f(x) = x < log(threshold) ? alpha*x : x when biased (distributions.py:125-134).
 * The contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32), one wave per 32x32 tile of node
 * pairs, the grid spans (sample, trace) x tiles.  max_node_count <= 1536, dim <= 80.
 * fwd workspace (tv_adj_bernoulli_ll_fwd_workspace_bytes): one partial sum per workgroup, added
 * per trace in a fixed order by a second kernel.
 * bwd: dh given d(ll); workspace (0 bytes for most shapes): with few traces of many nodes the
 * column blocks of a row block are split over several workgroups, whose partial dh tiles are
 * added in a fixed order. */
int64_t tv_adj_bernoulli_ll_fwd_workspace_bytes(int64_t n_samples, int64_t batch,
                                                int64_t max_node_count);
int tv_adj_bernoulli_ll_fwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                            int64_t dim, const float *h, const float *adj, int biased,
                            float alpha, float log_threshold, float *ll, void *workspace,
                            int64_t workspace_bytes, void *stream);
int64_t tv_adj_bernoulli_ll_bwd_workspace_bytes(int64_t n_samples, int64_t batch,
                                                int64_t max_node_count, int64_t dim);
int tv_adj_bernoulli_ll_bwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                            int64_t dim, const float *h, const float *adj, int biased,
                            float alpha, float log_threshold, const float *dll, float *dh,
                            void *workspace, int64_t workspace_bytes, void *stream);

/* The logits of p(adj|z) as a tensor: edge_logits_by_dot_product + dense_to_triu
 * (tensor_utils.py:54-67, model/struct_vae.py:281-282) for the callers that need the values
 * themselves - `make_graph` draws the adjacency from them when p() runs without an observed
 * graph (model/struct_vae.py:296-333), and `.distribution.logits` of the adj node.
 * logits[g, t] = <h[g,i,:], h[g,j,:]> for the t-th pair i < j in dense_to_triu order; fp32 fma
 * chain over dim.  graphs <= 65535 per call, max_node_count <= 1536. */
int tv_edge_logits_triu(int64_t graphs, int64_t max_node_count, int64_t dim, const float *h,
                        float *logits, void *stream);

/* No counterpart in the reference: tv_adj_bernoulli_ll_fwd for callers that cannot allocate
 * device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  Same inputs, outputs and results, no workspace argument: the partial sums live
 * in a device array the library holds on every GPU it runs on (TV_ADJ_STATIC_MAX_PARTIALS
 * floats = 4 MiB, until the process ends).  Calls whose n_samples * batch * tile groups
 * (= tv_adj_bernoulli_ll_fwd_workspace_bytes / 4) exceed that return TV_ERR_WORKSPACE.  The
 * call makes no hipMalloc / hipFree.  All calls on one GPU share the array: calls that overlap
 * on one GPU (different streams) race - serialise them. */
#define TV_ADJ_STATIC_MAX_PARTIALS (1 << 20)
int tv_adj_bernoulli_ll_fwd_static(int64_t n_samples, int64_t batch, int64_t max_node_count,
                                   int64_t dim, const float *h, const float *adj, int biased,
                                   float alpha, float log_threshold, float *ll, void *stream);

/* Categorical log-prob per row: ll[r] = log_softmax(logits[r,:])[target[r % target_rows]]
 * (tk.Categorical, model/struct_vae.py:258-266 and :413); optional biased variant.
 * fwd also returns the row log-sum-exp for the backward pass.
 * bwd: dlogits = dll[r] * g * (onehot - softmax), g = alpha where the biased branch was taken.
 * dlogits rows are dlogits_stride >= classes floats apart and the columns classes ..
 * dlogits_stride - 1 of every row are written as zeros: with a stride that is a multiple of 4
 * the gradient can go into tv_gemm_f32 as a 16-byte-readable operand whatever `classes` is. */
int tv_categorical_ll_fwd(int64_t rows, int64_t classes, int64_t target_rows,
                          const float *logits, const int64_t *target, int biased, float alpha,
                          float log_threshold, float *ll, float *lse /* [rows], nullable */,
                          void *stream);
int tv_categorical_ll_bwd(int64_t rows, int64_t classes, int64_t target_rows,
                          const float *logits, const int64_t *target, int biased, float alpha,
                          float log_threshold, const float *lse, const float *dll,
                          float *dlogits, int64_t dlogits_stride, void *stream);

/* z = mean + exp(logstd) * eps ; logq[k,b] = sum_d Normal(z; mean, logstd)
 * (tk.Normal sample + log_prob, model/struct_vae.py:238-241, model/latency_vae.py:267-270).
 * logstd is clamped to [lo, hi] first (T.maybe_clip, struct_vae.py:231-235).
 * mean/logstd_raw [B,D], eps [K,B,D] -> z [K,B,D], logq [K,B], logstd [B,D] (clamped). */
int tv_normal_reparam_fwd(int64_t n_samples, int64_t batch, int64_t dim, const float *mean,
                          const float *logstd_raw, float lo, float hi, const float *eps,
                          float *z, float *logq, float *logstd, void *stream);
int tv_normal_reparam_bwd(int64_t n_samples, int64_t batch, int64_t dim, const float *logstd_raw,
                          float lo, float hi, const float *eps, const float *dz,
                          const float *dlogq, float *dmean, float *dlogstd_raw, void *stream);
/* ll[r] = sum_d Normal(x[r,d]; mean[r,d], clamp(logstd_raw[r,d], lo, hi))  (p(z2|z),
 * model/latency_vae.py:284-311).  Gradients for x, mean, logstd_raw. */
int tv_normal_ll_fwd(int64_t rows, int64_t dim, const float *x, const float *mean,
                     const float *logstd_raw, float lo, float hi, float *ll, void *stream);
int tv_normal_ll_bwd(int64_t rows, int64_t dim, const float *x, const float *mean,
                     const float *logstd_raw, float lo, float hi, const float *dll, float *dx,
                     float *dmean, float *dlogstd_raw, void *stream);

/* log p(z) of the RealNVP prior of the struct VAE: `levels` x (ActNorm, affine coupling with ONE
 * hidden layer and sigmoid scale, invertible dense), then the unit normal
 * (model/realnvp_flow.py:18-49 via make_realnvp_flow(...).invert(), struct_vae.py:148-149,
 * 251-254; tensorkit ActNorm / CouplingLayer / InvertibleDense).  z [rows, dim] -> log_prob [rows].
 * `params` / `grads`: HOST arrays of levels*9 device pointers, per level in this order:
 *   0 act-norm bias [dim]         1 act-norm pre_scale [dim]
 *   2 hidden weight [hidden, n1]  3 hidden bias [hidden]          (n1 = dim/2, n2 = dim - n1)
 *   4 shift weight [n2, hidden]   5 shift bias [n2]
 *   6 pre-scale weight [n2, hidden]   7 pre-scale bias [n2]
 *   8 dense kernel [dim, dim]  (y = x K^T; log|det K| and K^-T are computed on the device)
 * A level maps x -> K [x1, (x2 + shift(x1)) * sigmoid(pre(x1) + 2)] with x = (x_in + bias) *
 * exp(pre_scale); LeakyReLU(slope) after the hidden layer.
 * ws: tv_realnvp_workspace_floats(rows, ...) floats; the backward call must get the workspace
 * its forward call filled (same parameters).  A NULL entry of `grads` skips that tensor.
 * tv_realnvp_supported: kernels exist for dim 10 / 16 / 20 with hidden 64. */
int tv_realnvp_supported(int dim, int hidden);
/* Data-dependent initialisation of the ActNorm layers on the first batch (tensorkit ActNorm;
 * the `initialize()` pass of train.py:188-199): level by level bias = -mean, pre_scale =
 * -log(max(std, 1e-6)) of the level's input over the rows (biased variance), written into
 * params[l*9 + 0] and params[l*9 + 1]; the other tensors are only read.  1 <= rows <= 2^24 (up to
 * 1024 rows a lane holds its row through the levels; larger batches recompute the level inputs). */
int tv_realnvp_actnorm_init(int64_t rows, int levels, int dim, int hidden, float slope,
                            const float *z, float *const *params, void *stream);
int64_t tv_realnvp_workspace_floats(int64_t rows, int levels, int dim, int hidden);
int tv_realnvp_logprob_fwd(int64_t rows, int levels, int dim, int hidden, float slope,
                           const float *z, const float *const *params, float *ws,
                           float *log_prob, void *stream);
int tv_realnvp_logprob_bwd(int64_t rows, int levels, int dim, int hidden, float slope,
                           const float *z, const float *const *params, const float *dlog_prob,
                           float *ws, float *dz, float *const *grads, void *stream);

/* p(latency|z2,G) (model/latency_vae.py:346-444, distributions.py:105-113, 267-286, 300-315,
 * 341-356): feats [K*B*M, 2] = (mean, raw logstd); logstd = max(raw, logstd_min), then
 * min(., log(max(std_limit[node_type_pad], 1e-7))) if std_limit != NULL;
 * mode 0 Normal, 1 SafeNormal(thr), 2 BiasedNormal(alpha, thr), 3 AnomalyDetectionNormal;
 * masked to m < node_count[b], weighted by M/node_count[b] when `weighted`.
 * ll [K,B]; optional std_out [K*B*M] (collect_latency_std, tensor_utils.py:155-169).
 * node_offsets != NULL: feats (and dfeats) are compact rows [K*(N+B), 2] (see tv_batch_build):
 * slot m < n_b reads row node_offsets[b]+m, a padded slot the representative row N+b;
 * needs use_mask. */
int tv_latency_ll_fwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                      const float *feats, const float *latency_pad, const int64_t *node_count,
                      const int64_t *node_type_pad, const int64_t *node_offsets,
                      const float *std_limit, int has_logstd_min, float logstd_min, int mode,
                      float std_threshold, float alpha, float log_bias_threshold, int use_mask,
                      int weighted, float *ll,
                      float *std_out, void *stream);
int tv_latency_ll_bwd(int64_t n_samples, int64_t batch, int64_t max_node_count,
                      const float *feats, const float *latency_pad, const int64_t *node_count,
                      const int64_t *node_type_pad, const int64_t *node_offsets,
                      const float *std_limit, int has_logstd_min, float logstd_min, int mode,
                      float std_threshold, float alpha, float log_bias_threshold, int use_mask,
                      int weighted, const float *dll,
                      float *dfeats, void *stream);

/* test-time scoring (evaluation.py:141-148): w = log_joint - latent_log_joint [K,B];
 * loss[b] = -mean_k w ; nll[b] = -(logsumexp_k w - log K) ; both clipped:
 * x < clip ? x : clip (NaN -> clip).  has_clip = 0 disables the clip. */
int tv_score_reduce(int64_t n_samples, int64_t batch, const float *w, int has_clip, float clip,
                    float *loss, float *nll, void *stream);

/* ------------------------------------------------------------------------------------
 * optimizer over one flat fp32 parameter buffer (train.py:369-372, 417-426: tk.optim
 * RMSprop / Adam after clip_grad_norm_(global 10)).
 *   tv_l2_sumsq_partial: partial[nblk] sums of p^2 over elements with l2_mask != 0
 *   tv_sumsq_partial   : partial[nblk] sums of g^2
 *   tv_optim_step      : scale = min(1, clip/(norm+1e-6)) read from device norm (or 1 if
 *                        clip_norm <= 0); kind 0 RMSprop(alpha .99, eps 1e-8), 1 Adam.
 * ---------------------------------------------------------------------------------- */
int64_t tv_reduce_blocks(int64_t n);
int tv_sumsq_partial(int64_t n, const float *x, const uint8_t *mask /* nullable */,
                     double *partial, void *stream);
int tv_sum_partials(int64_t nblk, const double *partial, int take_sqrt, float *out, void *stream);
/* y[i] += alpha * x[i] where mask[i] != 0 (gradient of the L2 term: alpha = 2*l2_reg) */
int tv_axpy_masked(int64_t n, float alpha, const float *x, const uint8_t *mask /* nullable */,
                   float *y, void *stream);
int tv_optim_step(int kind, int64_t n, float *param, const float *grad, float *state1,
                  float *state2, const float *grad_norm /* device scalar, nullable */,
                  float clip_norm, float lr, float beta1, float beta2, float eps, int64_t step,
                  void *stream);
/* tv_optim_step with the NaN decision of the step (train.py:272-274: a NaN loss raises before
 * optimizer.step()) taken ON THE DEVICE, so that the host does not have to wait for the loss of
 * every step: `guard` is a device scalar, and nothing is updated when it is NaN (guard_kind 1:
 * the loss itself) or non-zero (guard_kind 2: the all-reduced count of ranks whose loss was NaN).
 * The host reads the same scalar one step later (Trainer.finish) and raises then; parameters and optimizer
 * state are those from before the NaN step, as in the reference.  guard_host (nullable): one
 * float of PINNED HOST memory (hipHostMalloc; torch's pin_memory) that receives the scalar by a
 * store from the kernel - visible to the host once an event recorded behind the call has
 * completed -, so that no device-to-host copy is queued behind the step.  guard NULL /
 * guard_kind 0: tv_optim_step. */
int tv_optim_step_guarded(int kind, int64_t n, float *param, const float *grad, float *state1,
                          float *state2, const float *grad_norm /* device scalar, nullable */,
                          float clip_norm, float lr, float beta1, float beta2, float eps,
                          int64_t step, const float *guard /* device scalar, nullable */,
                          int guard_kind, float *guard_host /* pinned host scalar, nullable */,
                          void *stream);

/* ------------------------------------------------------------------------------------
 * the scalar end of train_step (tracegnn/models/trace_vae/train.py:229-270):
 *   recons = mean_b sum_{kind 0} vec[b];  kl = mean_b (sum_{kind 1} vec[b] - sum_{kind 2} vec[b])
 *   loss = beta * kl - recons + l2_reg * l2_value[0]
 *          + sum_t z_regs[t] * (mean_b m_b^2 + mean_b (v_b - 1)^2),
 *   (m_b, v_b) = mean / biased variance of zs[t][b, :]  (get_moments, tensor_utils.py:117-125;
 *   the z_unit_ball_reg terms of train.py:259-270).
 * vecs: n_vec device vectors [batch] (log p(x) of the observed variables: kind 0, log q of the
 * latents: kind 1, log p of the latents: kind 2); the tables themselves are host arrays.
 * out3 (device) = loss, recons, kl.  bwd: d(loss)/d(vecs[i]) into dvecs[i], d(loss)/d(zs[t])
 * (regulariser only) into dzs[t], scaled by dloss[0] (device); NULL entries are skipped.
 * ---------------------------------------------------------------------------------- */
int tv_train_loss_fwd(int64_t batch, int n_vec, const float *const *vecs, const int *kinds,
                      int n_z, const float *const *zs, const int64_t *z_dims,
                      const float *z_regs, const float *l2_value, float l2_reg, float beta,
                      float *out3, void *stream);
int tv_train_loss_bwd(int64_t batch, int n_vec, const float *const *vecs, const int *kinds,
                      float *const *dvecs, int n_z, const float *const *zs,
                      const int64_t *z_dims, const float *z_regs, float *const *dzs, float beta,
                      const float *dloss, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRACEVAE_HIP_H */
