/*
 * ultra_nbfnet.h -- C ABI of the dense layer epilogues around rspmm (libultra_amd.so).
 *
 * These replace the torch op chains of the reference layer on the hot path; operands are plain fp32
 * device pointers, kernels are enqueued on `stream` (hipStream_t as void*), status codes as in
 * ultra_rspmm.h.  Built for the shapes of the shipped ULTRA checkpoints (hidden dim 64); other shapes
 * return ULTRA_ERR_UNSUPPORTED and the host layer keeps using its generic path.
 */
#ifndef ULTRA_NBFNET_H
#define ULTRA_NBFNET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ULTRA_CONV_LAYER_NORM 1
#define ULTRA_CONV_RELU 2
#define ULTRA_CONV_RESIDUAL 4
#define ULTRA_LAYER0_MAX 8      /* ultra_nbf_layer0 only: max aggregate instead of sum (layers.py:206-207) */
#define ULTRA_LAYER0_ONLY_FILL 16   /* ... only the constant rows (they depend on the layer's parameters alone: relation, src_rows,
                                       src_values, weight may be NULL) -- e.g. on a side stream, beside the relation model */
#define ULTRA_LAYER0_SKIP_FILL 32   /* ... only the special rows, into an output the caller has filled with ULTRA_LAYER0_ONLY_FILL */

/*
 * GeneralizedRelationalConv.update (/root/reference/ultra/layers.py:233-240) fused with the residual
 * of the Bellman-Ford loop (/root/reference/ultra/models.py:158-160):
 *     out = [x +] relu( LayerNorm_eps( W . cat[x, agg] + b ) )
 * x, agg, out: (rows, 64) contiguous; weight (64, 128) row-major = linear.weight; bias (64) may be NULL;
 * ln_weight / ln_bias (64) required with ULTRA_CONV_LAYER_NORM.  out may not alias x or agg.
 */
int32_t ultra_conv_update(const void *x, const void *agg, const void *weight, const void *bias, const void *ln_weight,
                          const void *ln_bias, void *out, int64_t rows, int32_t input_dim, int32_t output_dim, float eps,
                          int32_t flags, void *stream);

/*
 * Backward of ultra_conv_update for the fine-tuning path (autograd of /root/reference/ultra/layers.py:233-240; in the
 * reference: cat, addmm, native_layer_norm, relu and add nodes).  Nothing but x and agg has to be kept from the forward:
 * the pre-activation is recomputed on the matrix cores.
 *     grad_x, grad_agg (rows, 64); grad_weight (64, 128); grad_bias / grad_ln_weight / grad_ln_bias (64) may be NULL.
 * `flags` as in the forward.  workspace: ultra_conv_update_backward_workspace(rows) bytes of device memory (the
 * pre-activation gradient and the per-workgroup partial sums; combined in a fixed order -- no atomics, gradients are
 * reproducible run to run).  Gradients OVERWRITE their destinations.
 */
int64_t ultra_conv_update_backward_workspace(int64_t rows);
int32_t ultra_conv_update_backward(const void *x, const void *agg, const void *grad_out, const void *weight, const void *bias,
                                   const void *ln_weight, const void *ln_bias, void *grad_x, void *grad_agg, void *grad_weight,
                                   void *grad_bias, void *grad_ln_weight, void *grad_ln_bias, void *workspace,
                                   int64_t workspace_bytes, int64_t rows, int32_t input_dim, int32_t output_dim, float eps,
                                   int32_t flags, void *stream);

/*
 * Readout of EntityNBFNet.forward (/root/reference/ultra/models.py:166-170, 202-209):
 *     feature = cat[hidden, query]; score = mlp.2( relu( mlp.0( feature.gather(t_index) ) ) )
 * in the reference's operation order: mlp.0 = one k-ascending fmaf chain per hidden unit over the 64 node features and
 * then the 64 query features, bias added after the chain (the concatenated feature is never materialised: the query half
 * of the chain reads query[sample]); mlp.2 = nn.Linear(128, 1), a GEMV on the reference's CPU path whose association of
 * the 128 products belongs to the host BLAS and is passed in as a PROGRAM (order_dev, at most 640 int32 words on the device;
 * ultra_amd/host_order.py recovers it from the BLAS of the running process):
 *     [n_stage, then one header per stage: L (lanes, power of two <= 16), carry (0/1), (offset_p, groups_p) for p < L,
 *      then the element area]: lane p's elements start at word offset_p (a multiple of 4) and fill groups_p groups of
 *     eight, the last one padded with the element 128, which multiplies zeros
 *     lane p: v = (p == 0 && carry) ? previous stage's result : 0;  v = fma(hid[k], w2[k], v) for its k in order
 *             (an element written k + 256: v = v + fl(hid[k] * w2[k]), for host code that does not fuse);
 *     fold: v[p] += v[p + L/2]; v[p] += v[p + L/4]; ...; stage result v[0];  score = last result + b2.
 * order_dev == NULL: one chain, k ascending.  Every product of every stage must appear exactly once.
 * hidden (batch, num_node, 64) contiguous; query (batch, 64); t_index (batch, n_cand) int64 node ids or NULL for
 * all-tail (n_cand == num_node, identity); w1 = mlp.0.weight (128, 128) row-major, b1 = mlp.0.bias (128);
 * w2 = mlp.2.weight (128); b2 = mlp.2.bias (1); score (batch, n_cand).
 */
int32_t ultra_readout(const void *hidden, const int64_t *t_index, const void *w1, const void *query, const void *b1,
                      const void *w2, const void *b2, const int32_t *order_dev, int64_t order_len, void *score, int64_t batch,
                      int64_t num_node, int64_t n_cand, int32_t hidden_dim, int32_t feature_dim, void *stream);

/*
 * Batch prologue of EntityNBFNet.forward (/root/reference/ultra/models.py:190-197 with
 * /root/reference/ultra/base_nbfnet.py:79-86) in one pass over triples = (batch, n_cand, 3) int64 [h, t, r]:
 *   side[b] = 1 if row b keeps its head fixed (tail candidates), 0 if it keeps its tail fixed (head candidates, turned
 *             into a tail query with the inverse relation r + num_direct_rel);
 *   h0[b], r0[b] = source node / query relation of row b after that conversion;
 *   valid[b] = 1 iff row b shares its source node and its relation (the reference's two asserts), else 0  (int32 [batch]).
 * rel_first (optional, int64 [batch]): row b's relation as given, triples[b, 0, 2] -- the relation model's query
 * (/root/reference/ultra/models.py:20).  One workgroup per row: no scratch, any number of prologues may run concurrently.
 * ultra_readout_batch is ultra_readout reading the candidate node straight from `triples` (column 1 or 0 by side[b]).
 */
int32_t ultra_batch_prologue(const int64_t *triples, int64_t batch, int64_t n_cand, int64_t num_direct_rel, int64_t *h0,
                             int64_t *r0, int32_t *side, int32_t *valid, int64_t *rel_first, void *stream);
/* ultra_batch_prologue with one more output for the training step (rows of a few hundred candidates): cand (batch, n_cand)
 * int64 = the candidate nodes of the converted rows, new_t_index of /root/reference/ultra/base_nbfnet.py:84 (the tail where
 * side[b] = 1, the head otherwise).  cand may be NULL. */
int32_t ultra_batch_prologue_rows(const int64_t *triples, int64_t batch, int64_t n_cand, int64_t num_direct_rel, int64_t *h0,
                                  int64_t *r0, int32_t *side, int32_t *valid, int64_t *rel_first, int64_t *cand, void *stream);
int32_t ultra_readout_batch(const void *hidden, const int64_t *triples, const int32_t *side, const void *w1, const void *query,
                            const void *b1, const void *w2, const void *b2, const int32_t *order_dev, int64_t order_len,
                            void *score, int64_t batch, int64_t num_node, int64_t n_cand, int32_t hidden_dim,
                            int32_t feature_dim, void *stream);

/*
 * The NBFNet boundary condition (/root/reference/ultra/models.py:59-66, 135-141): out (batch, num_node, dim) fp32,
 * out[b, n, :] = values[b, :] if n == rows[b] else 0; values (batch, dim) or NULL for all-ones (RelNBFNet's query).
 * One pass, no memset (hipGraph friendly).  dim a multiple of 4.
 */
int32_t ultra_onehot_rows(void *out, const int64_t *rows, const void *values, int64_t batch, int64_t num_node, int64_t dim,
                          void *stream);

/*
 * Dynamic edge dropout of the training step (/root/reference/ultra/base_nbfnet.py:54-77) as a 0/1 vector instead of a
 * filtered copy of the graph: keep[e] = 0 where edge e equals one of the listed (easy) edges, else 1.  Edges are compared
 * through the mixed-radix key of the reference's edge_match (tasks.py:7-39): key = (head * num_node + tail) * num_rel +
 * type, or head * num_node + tail when type == NULL (`remove_one_hop`).  easy_key_sorted: the n_easy <= 8192 keys of the
 * edges to drop, ascending (duplicates allowed).  keep: (num_edge) fp32.
 */
int32_t ultra_edge_keep_mask(const int64_t *head, const int64_t *tail, const int64_t *type, int64_t num_edge,
                             const int64_t *easy_key_sorted, int64_t n_easy, int64_t num_node, int64_t num_rel, void *keep,
                             void *stream);
/*
 * The same vector straight from the batch's triples (base_nbfnet.py:57-59 builds the list: every (h, t, r) of the batch and
 * its inverse (t, h, r + inverse_offset), inverse_offset = num_relations / 2): h, t, r point at the first of n_triple <= 4096
 * int64 ids each, `stride` elements apart (3 for the columns of a contiguous (batch, n, 3) tensor, 1 for separate vectors).
 * Every workgroup hashes the 2 n_triple keys into a table in LDS and looks each of its edges up: no list, no sort.
 * type == NULL / r == NULL (`remove_one_hop`): keys of (head, tail) alone.  Same keys, same result as ultra_edge_keep_mask.
 */
int32_t ultra_easy_edge_keep(const int64_t *head, const int64_t *tail, const int64_t *type, int64_t num_edge, const int64_t *h,
                             const int64_t *t, const int64_t *r, int64_t n_triple, int64_t stride, int64_t num_node,
                             int64_t num_rel, int64_t inverse_offset, void *keep, void *stream);

/*
 * Boundary condition of EntityNBFNet (/root/reference/ultra/models.py:131-141) with the query gather fused:
 *     query_out[b, :] = table[b, pick[b], :]            (query = relation_representations[arange(batch), r_index])
 *     out[b, n, :]    = query_out[b, :] if n == rows[b] else 0
 * table (batch, table_rows, dim) contiguous fp32; rows, pick int64 [batch]; dim a multiple of 4.
 * out may be NULL: then only query_out (and qbias_out) are produced and the boundary stays in closed form for
 * ultra_rspmm_forward_point / ultra_nbf_layer0.
 * Optionally (w1, b1, qbias_out all non-NULL) the same launch also emits the readout's per-sample bias
 *     qbias_out[b, f] = b1[f] + sum_k w1[f, dim + k] * query_out[b, k]      w1 (2 dim, 2 dim) = mlp.0.weight, f < 2 dim
 * (a per-sample folding of the readout's query half; the readout kernels of ABI 4 run the chain through the query
 * themselves and no longer take it).
 */
int32_t ultra_query_boundary(void *out, void *query_out, const int64_t *rows, const void *table, const int64_t *pick,
                             int64_t batch, int64_t num_node, int64_t table_rows, int64_t dim, const void *w1, const void *b1,
                             void *qbias_out, void *stream);

/*
 * The relation_projection MLPs of all entity layers (/root/reference/ultra/layers.py:80, applied to the relation
 * representations set by /root/reference/ultra/models.py:184-185) in one launch:
 *     out[l, r, :] = w2[l] . relu(w0[l] . x[r, :] + b0[l]) + b2[l]
 * x (rows, 64); w0, w2 (n_layer, 64, 64) = the stacked nn.Linear weights [out][in]; b0, b2 (n_layer, 64);
 * out (n_layer, rows, 64).  dim must be 64.
 */
int32_t ultra_relation_projection(const void *x, const void *w0, const void *b0, const void *w2, const void *b2, void *out,
                                  int64_t rows, int32_t n_layer, int32_t dim, void *stream);

/*
 * The same products with every layer's parameters where they live (n_layer host arrays of device pointers: w0[l], w2[l] (64, 64)
 * row-major [out][in]; b0[l], b2[l] (64)) -- the training step, whose parameters change every step, needs no stacked copies.
 */
int32_t ultra_relation_projection_layers(const void *x, const void *const *w0, const void *const *b0, const void *const *w2,
                                         const void *const *b2, void *out, int64_t rows, int32_t n_layer, int32_t dim, void *stream);

/*
 * Backward of ultra_relation_projection_layers for all layers in three launches (csrc/relproj_bwd.hip):
 *     grad_x (rows, 64) = sum_l ((grad_out[l] W2_l) * [h_l > 0]) W0_l,      h_l = relu(x W0_l^T + b0_l) recomputed
 *     grad_w2[l] = grad_out[l]^T h_l,  grad_b2[l] = column sums of grad_out[l],  grad_w0[l] = gh_l^T x,  grad_b0[l] = column sums of gh_l
 * grad_out: n_layer device pointers to (rows, 64) fp32; grad_w0 / grad_w2 stacked (n_layer, 64, 64), grad_b0 / grad_b2 (n_layer, 64),
 * all written in full.  workspace: ultra_relation_projection_backward_workspace(rows, n_layer) bytes.  dim == 64, n_layer <= 8.
 * Fixed summation order (no atomics).
 */
int64_t ultra_relation_projection_backward_workspace(int64_t rows, int32_t n_layer);
int32_t ultra_relation_projection_backward(const void *x, const void *const *w0, const void *const *b0, const void *const *w2,
                                           const void *const *grad_out, void *grad_x, void *grad_w0, void *grad_b0, void *grad_w2,
                                           void *grad_b2, void *workspace, int64_t workspace_bytes, int64_t rows, int32_t n_layer,
                                           int32_t dim, void *stream);

/*
 * Filtered ranking without the (batch, N) mask (/root/reference/ultra/tasks.py:94-141):
 *     rank[q] = 1 + #{t : t not in known(q) and score[q, pos[q]] <= score[q, t]}
 *     num_negative[q] = n_cand - |known(q)|
 * known(q) = known_index[known_ptr[q] : known_ptr[q + 1]]: the de-duplicated ids of the query's known true
 * answers INCLUDING pos[q] (strict_negative_mask zeroes both, tasks.py:108-111).  score (batch, n_cand) fp32,
 * everything else int64, all device pointers.  Ties count against the positive exactly like tasks.py:137.
 */
int32_t ultra_filtered_rank(const void *score, const int64_t *pos_index, const int64_t *known_ptr,
                            const int64_t *known_index, int64_t batch, int64_t n_cand, int64_t *rank_out,
                            int64_t *num_negative_out, void *stream);

/*
 * Strict negative sampling (/root/reference/ultra/tasks.py:42-76, strict = True) for n_query positives, n_draw negatives each,
 * without the (batch, N) masks of tasks.py:94-130 and without the host round trip of nonzero():
 *     out[q, d] = the floor(rand[q, d] * count[q])-th entity id, ascending, that is neither a known answer of query q nor its
 *                 positive;   count[q] = num_node - |known(q) u {positive[q]}|            (= candidate[index] of tasks.py:57-61)
 * known(q) is the slice of sorted_keys with (anchor[q] * num_relation + relation[q]) * num_node <= key < ... + num_node, where
 * sorted_keys holds the graph's DISTINCT (anchor, relation, answer) triples as ascending int64 keys
 * (anchor * num_relation + relation) * num_node + answer -- built once per graph by the caller (tails of (head, relation) for the
 * tail half of a batch, heads of (tail, relation) for the head half).  rand: fp32 uniform draws in [0, 1), the caller's
 * (torch.rand in the reference's order, so the sampled ids are the reference's).  All device pointers; ids int64.
 */
int32_t ultra_strict_negatives(const int64_t *sorted_keys_dev, int64_t n_key, const int64_t *anchor_dev, const int64_t *relation_dev,
                               const int64_t *positive_dev, const void *rand_dev, int64_t n_query, int64_t n_draw, int64_t num_node,
                               int64_t num_relation, int64_t *out_dev, void *stream);

/*
 * The fine-tuning step's loss and its gradient in one launch (/root/reference/script/run.py:66-77):
 *     target = [1, 0, ..., 0];  l = binary_cross_entropy_with_logits(pred, target, reduction = none)
 *     w[:, 0] = 1;  w[:, 1:] = softmax(pred[:, 1:] / temperature)  (temperature > 0; a constant, as under run.py's no_grad)
 *                              or uniform_weight (= 1 / num_negative; temperature == 0)
 *     loss = mean_b( sum_i l w / sum_i w );   grad = d loss / d pred
 * pred, grad (rows, n) fp32 row-major; loss one fp32.  rows <= 4096, n >= 2.  Sums run in a fixed order (reproducible).
 */
int32_t ultra_ranking_loss(const void *pred, int64_t rows, int64_t n, float temperature, float uniform_weight, void *loss,
                           void *grad, void *stream);

/*
 * The readout of a training step on the candidates' rows (/root/reference/ultra/models.py:202-207: score = mlp(cat[hidden,
 * query]), mlp = Linear(128, 128), ReLU, Linear(128, 1)) and its backward -- one launch forward, two backward:
 *     h = relu([hid[b, j] ; query[b]] w1^T + b1)          score[b, j] = h . w2 + b2
 * hid (batch, n, 64), query (batch, 64), w1 (128, 128) = mlp.0.weight, b1 (128), w2 (128) = mlp.2.weight, b2 (1); all fp32.
 * forward writes h (batch * n, 128) for the backward and score (batch, n).  backward takes grad_score (batch, n) and writes
 * grad_hid (batch, n, 64), grad_query (batch, 64), grad_w1 (128, 128), grad_b1 (128), grad_w2 (128), grad_b2 (1); work: scratch of
 * ultra_readout_train_backward_workspace(batch, n) bytes.  Every sum has a fixed order (reproducible run to run).
 */
int32_t ultra_readout_train_forward(const void *hid, const void *query, const void *w1, const void *b1, const void *w2,
                                    const void *b2, void *h, void *score, int64_t batch, int64_t n, void *stream);
int64_t ultra_readout_train_backward_workspace(int64_t batch, int64_t n);
int32_t ultra_readout_train_backward(const void *grad_score, const void *h, const void *hid, const void *query, const void *w1,
                                     const void *w2, void *grad_hid, void *grad_query, void *grad_w1, void *grad_b1, void *grad_w2,
                                     void *grad_b2, void *work, int64_t work_bytes, int64_t batch, int64_t n, void *stream);

/*
 * Relation graph of a knowledge graph (/root/reference/ultra/tasks.py:144-199) on the GPU, as bit matrices.
 *   edge_index (2, num_edge) int64 [head; tail], edge_type (num_edge) int64 -- inverse edges already included;
 *   W = (num_relation + 31) / 32 words per bit row.
 * ultra_relation_graph_bits:  hbits / tbits (num_node * W words, ZEROED by the caller) receive, per entity, the set of
 *   relations it is head / tail of; adj (4 * num_relation * W words, zeroed) the four adjacency bit matrices
 *   [hh | tt | ht | th] (row = r1, bit = r2; tasks.py:186-189); row_counts (4 * num_relation int64) the edges per (type, row).
 * ultra_relation_graph_emit:  with row_offsets = the exclusive prefix sum of row_counts (type-major, row-minor) and
 *   total_edges their sum, writes the relation graph's edge_index (2, total_edges) / edge_type (total_edges) in the
 *   reference's order (hh, tt, ht, th blocks, each sorted by (row, col)).
 * ultra_relation_graph_dense_adjacency:  the same matrices as the byte adjacency of the reference-order layer kernel
 *   (ultra_nbf_dense_layer with ULTRA_LAYER_REFERENCE_ORDER): a_ex_out = ceil(R/16)^2 * 1024 bytes, layout
 *   [row tile 16][col chunk 16][lane = row % 16 + 16 type][col % 16] -- plan format straight from the device.
 */
int32_t ultra_relation_graph_bits(const int64_t *edge_index_dev, const int64_t *edge_type_dev, int64_t num_edge, int64_t num_node,
                                  int64_t num_relation, void *hbits_dev, void *tbits_dev, void *adj_dev, int64_t *row_counts_dev,
                                  void *stream);
int32_t ultra_relation_graph_emit(const void *adj_dev, const int64_t *row_offsets_dev, int64_t num_relation, int64_t total_edges,
                                  int64_t *edge_index_out_dev, int64_t *edge_type_out_dev, void *stream);
int32_t ultra_relation_graph_dense_adjacency(const void *adj_dev, int64_t num_relation, void *a_ex_out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ULTRA_NBFNET_H */
