/*
 * ultra_rspmm.h -- C ABI of the MI355X-native relational SpMM engine (libultra_amd.so).
 *
 * This is the drop-in boundary for ULTRA's one native component:
 *   /root/reference/ultra/rspmm/source/rspmm.h:63-105   (the 12 rspmm_<sum>_<mul>_{forward,backward}_cuda exports)
 *   /root/reference/ultra/rspmm/source/rspmm.cpp:270-282 (their pybind registration)
 *   /root/reference/ultra/rspmm/rspmm.py:168-179         (generalized_rspmm, the Python dispatcher above them)
 *
 * Semantics (rspmm.cpp:50-75, operator.cuh:13-80):
 *   out[row, d] = NARY_{e : edge_index[0][e] == row}  w[e] * BINARY(rel[edge_type[e], d], in[edge_index[1][e], d])
 *   NARY in {add, min, max} with identity 0 / +MAX / -MAX (empty rows keep the identity),
 *   BINARY in {mul (DistMult), add (TransE)};  dtype float32 or float64.
 *
 * Plain pointers and sizes only -- no torch types.  All `*_dev` pointers are device (HBM) addresses
 * on the current HIP device, `*_host` pointers are host addresses.  Every function returns 0 on
 * success and a non-zero ultra_status otherwise; ultra_last_error() returns the message of the last
 * failure on the calling thread.  Kernels are enqueued on `stream` (a hipStream_t passed as void*,
 * NULL = the default stream) and never synchronise it.
 *
 * Two layers:
 *   1. plan API (fast path).  The graph is static across the 12 rspmm calls of an Ultra.forward and
 *      across batches, so the per-call argsort + ind2ptr of the reference (rspmm.py:175-177,
 *      rspmm.cpp:40-48) is hoisted into a plan built once per graph.
 *   2. reference-shaped stateless entry points ultra_rspmm_<sum>_<mul>_{forward,backward}_cuda with
 *      exactly the reference exports' operands (they build a throw-away plan per call).
 */
#ifndef ULTRA_RSPMM_H
#define ULTRA_RSPMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ULTRA_ABI_VERSION 7

typedef enum {
    ULTRA_OK = 0,
    ULTRA_ERR_INVALID = 1,      /* bad argument (the reference's TORCH_CHECK / c10::Error cases, rspmm.cpp:15-38) */
    ULTRA_ERR_UNSORTED = 2,     /* stateless entry got unsorted edge_index (rspmm.py:18 AssertionError) */
    ULTRA_ERR_HIP = 3,          /* a HIP runtime call failed (no GPU, OOM, launch failure) */
    ULTRA_ERR_UNSUPPORTED = 4
} ultra_status;

typedef enum { ULTRA_SUM_ADD = 0, ULTRA_SUM_MIN = 1, ULTRA_SUM_MAX = 2 } ultra_sum;   /* operator.cuh:43-80 */
typedef enum { ULTRA_MUL_MUL = 0, ULTRA_MUL_ADD = 1 } ultra_mul;                     /* operator.cuh:13-41 */
typedef enum { ULTRA_F32 = 0, ULTRA_F64 = 1 } ultra_dtype;                           /* AT_DISPATCH_FLOATING_TYPES, rspmm.cpp:148 */

/*
 * Dense operand: logically [n_outer][n_row][row_len], element (o, r, d) at
 * ptr[o * stride_outer + r * stride_row + d] (strides in ELEMENTS, innermost contiguous).
 *   reference layout  (N, D) row-major, D = batch * dim  (layers.py:190-192):  n_outer = 1, row_len = D
 *   batch-major layout (batch, N, dim), the module-level layout (models.py:139) : n_outer = batch, row_len = dim
 * A matrix shared by every outer slice (RelNBFNet's relation.weight.expand, layers.py:76) has
 * stride_outer = 0.  All operands of one call must agree on n_outer and row_len.
 */
typedef struct {
    void *ptr;
    int64_t n_outer;
    int64_t stride_outer;
    int64_t n_row;
    int64_t stride_row;
    int64_t row_len;
} ultra_mat;

/* Plan build options; zero-initialise for defaults. */
typedef struct {
    int32_t seg_len;   /* rows with more edges are split into segments of this many edges (0 -> 256);
                          ULTRA_PLAN_EXACT_ORDER: rows with more edges become chain rows (never split) */
    int32_t g_max;     /* rows with <= g_max edges are walked by one 16-lane group, longer ones by a whole wave (0 -> 64) */
    int32_t flags;     /* ULTRA_PLAN_* */
    int32_t reserved;
} ultra_plan_opts;

#define ULTRA_PLAN_EXACT_ORDER 1   /* the reference's summation order (rspmm.cpp:61-72): every row is summed sequentially in sorted
                                      (row, col, edge id) order -- by one 16-lane group, or for rows longer than seg_len by a
                                      workgroup-wide producer / consumer chain -- so sums equal the reference's bit for bit.
                                      No partial slots, no fix-up launch, no scratch: such a plan is immutable after upload and
                                      may be shared by concurrent streams. */
#define ULTRA_PLAN_TYPE_RUNS 2     /* edges sorted by (row, type, col) and cut at type changes: every item holds ONE relation, so
                                      add_mul sums the sources first and multiplies by rel[type] once per item.  Pays off when runs are
                                      long (dense graphs with few relation types, e.g. ULTRA's relation graph); add_mul only. */
#define ULTRA_PLAN_DENSE 4         /* dense-format plan for graphs that are (nearly) complete: the edge multiplicities (<= 255) are stored as
                                      num_relation dense (num_out_row x num_in_row) byte matrices in MFMA fragment order and
                                      add_mul runs as out = sum_t rel[t] * (A_t . x) on the matrix cores.  Serves fp32 add_mul with
                                      unit edge weights and row_len % 32 == 0 only (anything else: ULTRA_ERR_UNSUPPORTED);
                                      num_in_row <= ULTRA_DENSE_MAX_IN_ROW.  ULTRA's relation graphs (a few hundred nodes, 4 types,
                                      most (row, type, col) cells occupied) are the case it exists for. */
#define ULTRA_DENSE_MAX_IN_ROW 1024

typedef struct ultra_plan ultra_plan;

typedef struct {
    int64_t num_edge, num_node, num_relation;
    int64_t n_item, n_wave_item, n_group_item, n_unit;
    int64_t n_split_row, n_partial_slot;
    int32_t seg_len, g_max, flags, packed;   /* packed: col/type share one 32-bit word */
    int32_t on_device;
    int32_t has_transpose;
    int64_t n_type_run;   /* number of distinct (row, type) pairs: num_edge / n_type_run = mean run length */
    int64_t dense_bytes;  /* ULTRA_PLAN_DENSE: size of the fragment-ordered adjacency, else 0 */
    int64_t n_chain_row;  /* ULTRA_PLAN_EXACT_ORDER: rows longer than seg_len, walked by a whole workgroup (items[0, n_chain_row)) */
    int64_t dense_order_bytes;  /* ULTRA_PLAN_DENSE: size of the adjacency of the reference-order layer kernel; 0 when the graph does
                                   not qualify (more than 4 relation types, repeated edges, parallel edges not sorted by type) */
} ultra_plan_info;

int32_t ultra_abi_version(void);
const char *ultra_last_error(void);

/*
 * Errors raised ON the device.  The waits of the one-launch layer's hand-off between walking and multiplying waves
 * (ultra_rspmm_forward_update, forms beside the walk) are bounded: a wave that has polled for >= 0.05 s stores an error word in
 * pinned host memory, lets the other side through and ends -- a protocol error or a broken schedule is a failed call, not a hung
 * GPU.  The word is looked at (and cleared) at the entry of every rspmm forward call and by this function: returns ULTRA_OK, or
 * ULTRA_ERR_HIP with ultra_last_error() naming the wait and the workgroup.  Call it after synchronising the stream to learn
 * about the launches before; no HIP call is made.
 */
int32_t ultra_device_error(void);

/* Number of GPUs the HIP runtime sees (0 without a GPU; never fails). */
int32_t ultra_device_count(void);

/*
 * Build the aggregation plan of a graph on the host (no HIP calls).
 *   edge_index_host: (2, num_edge) int64 row-major, [0] = aggregation target, [1] = gathered source (rspmm.cpp:143-145)
 *   edge_type_host : (num_edge) int64 in [0, num_relation)
 * Edges may come in any order (generalized_rspmm accepts unsorted input, rspmm.py:175-179).
 * num_out_row rows are produced, sources index [0, num_in_row); the reference always has
 * num_out_row == num_in_row == input.size(0) (rspmm.cpp:139).
 */
int32_t ultra_plan_create(ultra_plan **plan, const int64_t *edge_index_host, const int64_t *edge_type_host,
                          int64_t num_edge, int64_t num_out_row, int64_t num_in_row, int64_t num_relation,
                          const ultra_plan_opts *opts);
/* Copy the plan arrays to the current HIP device (idempotent). */
int32_t ultra_plan_upload(ultra_plan *plan);
int32_t ultra_plan_destroy(ultra_plan *plan);
/* delta = +1 / -1: a captured hipGraph starts / stops referencing this plan's device arrays.  While pinned, a call that
 * would have to re-allocate one of the plan's scratch buffers (general-walk plans only: per-call weights in sorted order,
 * partial sums of split rows) fails with ULTRA_ERR_INVALID instead of freeing memory the graph still reads.
 * ULTRA_PLAN_EXACT_ORDER plans own no scratch. */
int32_t ultra_plan_pin(ultra_plan *plan, int32_t delta);
int32_t ultra_plan_get_info(const ultra_plan *plan, ultra_plan_info *info);

/* Host-side introspection (tests, tooling): copies array `which` into dst, returns element count via *count. */
typedef enum {
    ULTRA_ARR_ROW_PTR = 0,   /* int32 [num_out_row + 1] */
    ULTRA_ARR_COL = 1,       /* int32 [num_edge] sorted order */
    ULTRA_ARR_TYPE = 2,      /* int32 [num_edge] */
    ULTRA_ARR_PERM = 3,      /* int32 [num_edge]  sorted position -> original edge id */
    ULTRA_ARR_ITEM = 4,      /* int32 [n_item][4] = {row, begin, len, slot}  (slot < 0: writes the output row directly) */
    ULTRA_ARR_SPLIT_ROW = 5, /* int32 [n_split_row] */
    ULTRA_ARR_SPLIT_PTR = 6, /* int32 [n_split_row + 1] partial-slot ranges */
    ULTRA_ARR_DENSE = 7      /* uint8 [row_tile][type_chunk][kgroup (padded to a multiple of 20)][lane 0..63][tl 0..tc-1][q 0..3], in 4-byte words
                                (ULTRA_PLAN_DENSE only; tc = 1, 2, 4 for num_relation 1, 2, >= 3): multiplicity of edge
                                (row = 32 row_tile + lane % 32, type = tc type_chunk + tl, col = 8 kgroup + 2 q + lane / 32) */
    , ULTRA_ARR_DENSE_ORDER = 8   /* uint8 [row_tile16][col_chunk16][lane = row % 16 + 16 type][col % 16], in 4-byte words: the 0 / 1
                                     adjacency of the reference-order layer kernel (ULTRA_PLAN_DENSE plans with dense_order_bytes > 0) */
} ultra_plan_array;
int32_t ultra_plan_export(const ultra_plan *plan, int32_t which, void *dst_host, int64_t capacity_elems, int64_t *count);

/*
 * Forward.  output = NARY-aggregate (+ fused boundary when boundary != NULL):
 *   boundary fused with the same NARY op -- layers.py:199-207: sum -> update + boundary, max -> max(update, boundary).
 * edge_weight_dev: num_edge values of `dtype` in ORIGINAL edge order, or NULL for all-ones
 * (the fused path of the reference always passes ones, models.py:143).
 */
int32_t ultra_rspmm_forward(ultra_plan *plan, int32_t sum, int32_t mul, int32_t dtype,
                            const void *edge_weight_dev, const ultra_mat *relation, const ultra_mat *input,
                            const ultra_mat *boundary, const ultra_mat *output, void *stream);

/*
 * ultra_rspmm_forward with the weight stream read as a 0/1 KEEP MASK (dtype-typed, original edge order): an edge
 * with keep == 0 is absent from the graph for this call.  Under `add` that equals a zero weight; under `min` / `max` a
 * zero weight would still enter the reduction with the value 0, an absent edge does not.  This is the reference's
 * training-time edge dropout (/root/reference/ultra/base_nbfnet.py:54-77) without a filtered copy of the graph: the plan
 * of the static graph serves every batch.  ultra_rspmm_backward needs no twin: with 0/1 weights its min / max rule
 * (operator.cuh:62-64) already gives dropped edges a zero gradient.
 */
int32_t ultra_rspmm_forward_masked(ultra_plan *plan, int32_t sum, int32_t mul, int32_t dtype, const void *edge_keep_dev,
                                   const ultra_mat *relation, const ultra_mat *input, const ultra_mat *boundary,
                                   const ultra_mat *output, void *stream);

/*
 * Forward with a POINT boundary: the NBFNet boundary condition (/root/reference/ultra/models.py:59-66, 135-141) is zero
 * except for one row per outer slice, so `update + boundary` (/root/reference/ultra/layers.py:199-200) only touches that
 * row.  point_values: (n_outer, 1, row_len) -- n_row == 1 -- is added to output row point_rows[outer]; nothing of size
 * (n_outer, num_node, row_len) is read.
 * sum = ULTRA_SUM_MIN / _MAX (layers.py:206-207: max(update, boundary)): zero is not the identity of min / max, so the
 * boundary tensor the point stands for takes part at EVERY row -- row point_rows[outer] meets its value, every other row
 * meets 0 (an edge-less row therefore outputs 0, like the reference).  Served by ULTRA_PLAN_EXACT_ORDER plans in the
 * sparse format; ULTRA_ERR_UNSUPPORTED otherwise (the caller then passes the boundary as a tensor).
 */
int32_t ultra_rspmm_forward_point(ultra_plan *plan, int32_t sum, int32_t mul, int32_t dtype, const void *edge_weight_dev,
                                  const ultra_mat *relation, const ultra_mat *input, const int64_t *point_rows_dev,
                                  const ultra_mat *point_values, const ultra_mat *output, void *stream);

/*
 * Aggregate + layer update in ONE launch (fp32 inference path of GeneralizedRelationalConv.forward,
 * /root/reference/ultra/layers.py:84-131, 190-240, with the residual of /root/reference/ultra/models.py:158-160):
 *
 *   aggregate = rspmm(sum, mul)(relation, input) [+ point boundary]        -- as ultra_rspmm_forward_point / _forward
 *   output    = [input +] relu( LayerNorm( W . [input ; aggregate] + b ) ) -- as ultra_conv_update (ultra_nbfnet.h), same `flags` / `eps`
 *
 * Every workgroup of the reference-order kernel applies the update to the rows it aggregates, in one of two forms
 * (ultra_tuning.reserved[2]; 0 = the library chooses: form 3 where it fits and the graph has 10+ steps -- edges + rows --
 * a row, the tail form otherwise):
 *   1  in the kernel's tail, once its walks have ended;
 *   3  beside the walk, rows through LDS: twelve of the sixteen waves walk the graph and park every finished aggregate row
 *      and its input row in a ring in LDS, the four update waves (the weight matrix split over their registers) take them
 *      from there -- no aggregate ever travels through memory.  Every wait of that hand-off is bounded (ultra_device_error).
 *   (2 -- rows handed over by reference through memory -- existed in ABI 5 and was removed in ABI 6: ULTRA_ERR_UNSUPPORTED.)
 * Results are bit-equal with the two separate calls in either form.  sum: ULTRA_SUM_ADD / _MIN / _MAX (min / max with a
 * point boundary: see ultra_rspmm_forward_point).  Form 3 asked for explicitly where it does not fit the call (LDS, chain
 * rows per workgroup, a row pitch other than 256 bytes) is ULTRA_ERR_UNSUPPORTED.
 * `aggregate` is scratch for the caller (form 1 leaves the aggregate there; form 3 writes only the rows of its chains);
 * `output` must not alias it or `input`.
 * point_rows_dev / point_values: both NULL = no boundary.  Served where the stream walk serves ultra_rspmm_forward_point
 * (ULTRA_PLAN_EXACT_ORDER plan in the sparse format, 64-element rows, every stride equal): ULTRA_ERR_UNSUPPORTED
 * otherwise, nothing launched -- the caller then makes the two calls.
 */
int32_t ultra_rspmm_forward_update(ultra_plan *plan, int32_t sum, int32_t mul, const ultra_mat *relation, const ultra_mat *input,
                                   const int64_t *point_rows_dev, const ultra_mat *point_values, const ultra_mat *aggregate,
                                   const void *weight, const void *bias, const void *ln_weight, const void *ln_bias, float eps,
                                   int32_t flags, const ultra_mat *output, void *stream);

/*
 * Layer 0 of an NBFNet applied to its own boundary condition (/root/reference/ultra/models.py:72-80, 150-163 with
 * /root/reference/ultra/layers.py:183-207, 233-240; sum aggregate, DistMult message, hidden dim 64, fp32):
 *     x0[b, n] = src_values[b] (or ones if NULL) at n == src_rows[b], else 0
 *     out = [x0 +] relu( LayerNorm_eps( weight . cat[x0, rspmm(x0) + x0] + bias ) )        flags: ULTRA_CONV_* of ultra_nbfnet.h
 * With ULTRA_LAYER0_MAX the aggregate is max(rspmm_max(x0), x0) (layers.py:206-207) instead of the sum: the zero rows send
 * exact zeros and the boundary tensor is zero off the source row, so the rows not reached keep the same constant output.
 * Rows that are neither src_rows[b] nor a target of one of its out-edges all equal relu(LayerNorm(bias)); they are
 * filled, the others are computed from the transposed plan.  relation: (n_outer, num_relation, 64); weight (64, 128)
 * row-major = linear.weight; output (n_outer, num_node, 64).  Needs the (row, col) plan of a square graph.
 */
int32_t ultra_nbf_layer0(ultra_plan *plan, const void *edge_weight_dev, const ultra_mat *relation, const int64_t *src_rows_dev,
                         const void *src_values_dev, const void *weight, const void *bias, const void *ln_weight,
                         const void *ln_bias, float eps, int32_t flags, const ultra_mat *output, void *stream);

/*
 * One whole layer on a ULTRA_PLAN_DENSE plan (at most 4 relation types, square graph, hidden dim 64, fp32) in one launch:
 *     agg = rspmm add_mul (unit weights) + boundary;   out = [input +] relu( LayerNorm_eps( weight . cat[input, agg] + bias ) )
 * i.e. /root/reference/ultra/layers.py:183-207 + 233-240 with the residual of /root/reference/ultra/models.py:158-160 --
 * the steady-state layer of RelNBFNet.  boundary: NULL, a (n_outer, num_node, 64) tensor (point_rows_dev == NULL) or a
 * point boundary (n_outer, 1, 64) with point_rows_dev.  flags: ULTRA_CONV_* of ultra_nbfnet.h, plus
 * ULTRA_LAYER_REFERENCE_ORDER: the aggregate is summed in the reference's order (sorted edge order of every row,
 * /root/reference/ultra/rspmm/source/rspmm.cpp:61-72) -- bit-identical to the reference's rspmm + boundary; needs
 * dense_order_bytes > 0.  The aggregate never leaves the chip.  ULTRA_ERR_UNSUPPORTED when the plan / shapes do not fit
 * (callers then run rspmm + conv_update).
 */
#define ULTRA_LAYER_REFERENCE_ORDER 8
int32_t ultra_nbf_dense_layer(ultra_plan *plan, const ultra_mat *relation, const ultra_mat *input, const ultra_mat *boundary,
                              const int64_t *point_rows_dev, const void *weight, const void *bias, const void *ln_weight,
                              const void *ln_bias, float eps, int32_t flags, const ultra_mat *output, void *stream);

/*
 * add_mul forward for a ROW-SPARSE input: input[o] is zero outside row src_rows_dev[o] (int64, one per outer
 * slice) -- the layer-0 input of every NBFNet, whose boundary condition puts the query vector at the head node
 * and zeros elsewhere (/root/reference/ultra/models.py:59-66, 135-141).  Zero rows contribute exact zeros to a
 * sum of products, so only the edges gathered from the source row are visited (through the transposed plan, in
 * edge-id order per target: deterministic); the result equals ultra_rspmm_forward(add, mul) on the same
 * operands.  output is fully written (zero fill + contributions + boundary at the source row).
 * The CALLER guarantees the sparsity pattern; square graphs only.
 */
int32_t ultra_rspmm_forward_onehot(ultra_plan *plan, int32_t dtype, const void *edge_weight_dev,
                                   const ultra_mat *relation, const ultra_mat *input, const int64_t *src_rows_dev,
                                   const ultra_mat *boundary, const ultra_mat *output, void *stream);

/*
 * relation_grad of add_mul with unit edge weights on a ULTRA_PLAN_DENSE plan (/root/reference/ultra/rspmm/source/rspmm.cpp:106-108):
 *     relation_grad[o, t, :] = sum_{e : type_e == t} output_grad[o, row_e, :] * input[o, col_e, :]
 * as the per-type products A_t . input of the dense forward, weighed with output_grad and summed over the rows -- the backward of
 * ULTRA's relation graph (474 nodes, 4 types, 0.9 M edges) without a walk over its edge list.  fp32, row_len a multiple of 32,
 * at most 32 relation types, 16-byte aligned operands; otherwise ULTRA_ERR_UNSUPPORTED (use ultra_rspmm_backward).
 * Deterministic (fixed summation order, no atomics).
 */
int32_t ultra_rspmm_dense_relation_grad(ultra_plan *plan, const ultra_mat *input, const ultra_mat *output_grad,
                                        const ultra_mat *relation_grad, void *stream);

/*
 * add-aggregate rspmm restricted to a LIST of output rows per outer slice (fine-tuning: a training step reads the last
 * layer's output at its 1 + num_negative candidate rows only, /root/reference/ultra/models.py:202-207):
 *   forward   aggregate[o, j] = sum_{e : row_e == rows[o, j]} w_e * BINARY(rel[o, type_e], input[o, col_e])
 *                               + boundary[o, rows[o, j]]  (tensor, may be NULL)  + point_values[o] where rows[o, j] == point_rows[o]
 *   backward  input_grad[o, col_e] += w_e * dBINARY/dinput * aggregate_grad[o, j];  relation_grad[o, type_e] += w_e * dBINARY/drel * ...
 * rows: (n_outer, n_list) int64, repeats allowed; aggregate / aggregate_grad: (n_outer, n_list, row_len) fp32 contiguous.  The
 * backward ADDS into relation_grad / input_grad (the caller zeroes or pre-fills them) with float atomics: like the reference's
 * GPU backward (rspmm.cu:153-214) its last bits vary run to run.  fp32, row_len a multiple of 64, (row, col) plans;
 * edge_weight in original edge order (NULL = ones).  Anything else: ULTRA_ERR_UNSUPPORTED.
 */
int32_t ultra_rspmm_rows_forward(ultra_plan *plan, int32_t mul, const void *edge_weight_dev, const ultra_mat *relation,
                                 const ultra_mat *input, const int64_t *rows_dev, int64_t n_list, const ultra_mat *boundary,
                                 const int64_t *point_rows_dev, const void *point_values_dev, void *aggregate_dev, void *stream);
int32_t ultra_rspmm_rows_backward(ultra_plan *plan, int32_t mul, const void *edge_weight_dev, const ultra_mat *relation,
                                  const ultra_mat *input, const int64_t *rows_dev, int64_t n_list, const void *aggregate_grad_dev,
                                  const ultra_mat *relation_grad, const ultra_mat *input_grad, void *stream);

/*
 * The same backward as GATHERS -- no atomics, a fixed summation order, both gradients WRITTEN in full (no zeroing by the caller):
 *     input_grad[o, c]    = sum_{j : rows[o, j] == c} update_grad[o, j]  +  sum_{e : col_e == c, row_e listed} w_e dBINARY/dinput * aggregate_grad[o, j(row_e)]
 *     relation_grad[o, t] = sum_{e : type_e == t, row_e listed} w_e dBINARY/drel * aggregate_grad[o, j(row_e)]
 * (repeated list entries count once each, as in the scatter).  update_grad (n_outer, n_list, row_len; may be NULL): the share of
 * the input gradient that reaches the listed rows directly (the layer update reads input[o, rows[o, j]]).  point_values_grad
 * (n_outer, row_len; may be NULL) receives sum_{j : rows[o, j] == point_rows[o]} aggregate_grad[o, j] -- the gradient of the point
 * boundary's values.  Every destination row is summed by one owner that walks ITS edges (the edge list grouped by source / by type,
 * built with the plan on first use) and looks every edge's aggregation row up in a per-step table of the listed rows.
 * fp32, row_len == 64, n_list <= 1024, square graphs, plans that kept their edge list; otherwise ULTRA_ERR_UNSUPPORTED (use
 * ultra_rspmm_rows_backward).  Uses plan-owned scratch: one call at a time per plan.
 */
int32_t ultra_rspmm_rows_backward_gather(ultra_plan *plan, int32_t mul, const void *edge_weight_dev, const ultra_mat *relation,
                                         const ultra_mat *input, const int64_t *rows_dev, int64_t n_list,
                                         const void *aggregate_grad_dev, const void *update_grad_dev, const int64_t *point_rows_dev,
                                         void *point_values_grad_dev, const ultra_mat *relation_grad, const ultra_mat *input_grad,
                                         void *stream);

/*
 * Tag of the edge-weight vector passed to the NEXT weighted call of this thread (ultra_rspmm_forward / _masked / _point /
 * _backward); 0 = none.  Plans bring per-call weights (original edge order) into their own order with one small kernel per
 * call; a caller that hands the SAME vector to many calls -- a training step's 0/1 keep mask goes to every layer's forward and
 * backward walks -- tags it, and a plan that already holds the permutation of (this tag, this address, this stream) reuses it.
 * The caller promises: equal tag + equal address = equal contents.  Never used while the stream is capturing.
 */
int32_t ultra_rspmm_weight_epoch(int64_t epoch);

/*
 * Backward of ultra_rspmm_forward_onehot in its layer-0 use (fine-tuning): the input is the boundary condition -- values[o]
 * at row src_rows[o] of outer slice o, zero elsewhere -- and the same tensor is the boundary added to the sum.  Only the
 * edges leaving the source rows matter (rspmm.cpp:106-112 restricted to them):
 *   S[o, t]             = sum over edges e with source src_rows[o] and type t of  w_e * output_grad[o, target_e]
 *   relation_grad[o, t] = values[o] * S[o, t]                                   ((n_outer, num_relation, row_len) contiguous)
 *   values_grad[o]      = sum_t relation[o, t] * S[o, t] + output_grad[o, src_rows[o]]         ((n_outer, row_len) contiguous)
 * The graph arrives as the caller's CSR over SOURCE nodes with every node's edges sorted by type: out_ptr (num_node + 1),
 * out_edge (edge ids in (source, type) order), and the edge list's target / type arrays (original order; edge_weight too,
 * NULL = ones).  fp32, row_len a multiple of 64, 16-byte aligned rows, num_relation * 256 B + 34 KB <= 160 KB of LDS;
 * anything else: ULTRA_ERR_UNSUPPORTED.  No atomics: the same bits run to run.  Either gradient pointer may be NULL.
 */
int32_t ultra_rspmm_onehot_backward(const int64_t *out_ptr_dev, const int64_t *out_edge_dev, const int64_t *edge_target_dev,
                                    const int64_t *edge_type_dev, const void *edge_weight_dev, const ultra_mat *relation,
                                    const void *values_dev, const int64_t *src_rows_dev, const ultra_mat *output_grad,
                                    void *relation_grad_dev, void *values_grad_dev, void *stream);

/*
 * Backward (rspmm.cpp:77-119 / 164-219): gradients w.r.t. edge_weight (original edge order, may be
 * NULL to skip), relation and input, given the forward output and its gradient.  min/max give the
 * full gradient to every tying edge (operator.cuh:62-64,75-77).
 * relation_grad / input_grad are overwritten (the reference returns fresh zeros_like + accumulation).
 * input_grad may be NULL under sum == add: the input gradient is then left to the caller -- it is an rspmm forward over
 * the transposed graph (rspmm.cpp:110-112), which a graph with a dense-format twin runs on the matrix cores instead.
 */
int32_t ultra_rspmm_backward(ultra_plan *plan, int32_t sum, int32_t mul, int32_t dtype,
                             const void *edge_weight_dev, const ultra_mat *relation, const ultra_mat *input,
                             const ultra_mat *output, const ultra_mat *output_grad,
                             void *weight_grad_dev, const ultra_mat *relation_grad, const ultra_mat *input_grad,
                             void *stream);

/*
 * ultra_rspmm_backward under sum == add with input_grad = input_grad_base + (the gathered input gradient): the input of an
 * NBFNet layer feeds the rspmm AND the layer update, and the update's share of its gradient is added in the epilogue of the
 * walk instead of by a pass of its own over three (batch, N, d) tensors.  input_grad_base may alias input_grad.
 */
int32_t ultra_rspmm_backward_add(ultra_plan *plan, int32_t sum, int32_t mul, int32_t dtype,
                                 const void *edge_weight_dev, const ultra_mat *relation, const ultra_mat *input,
                                 const ultra_mat *output, const ultra_mat *output_grad,
                                 void *weight_grad_dev, const ultra_mat *relation_grad,
                                 const ultra_mat *input_grad_base, const ultra_mat *input_grad, void *stream);

/*
 * Reference-shaped stateless entry points (one per export of rspmm.h:63-105).  Operands are the
 * reference's: SORTED edge_index (2, E) int64, edge_type (E) int64, edge_weight (E), relation (R, D),
 * input (N, D), all contiguous device arrays; output (N, D) is written.  Unsorted edge_index ->
 * ULTRA_ERR_UNSORTED (rspmm.py:18).  They copy the edge list to the host to build a throw-away plan,
 * i.e. they synchronise -- like the reference wrapper does (rspmm.py:17-18,176).
 */
#define ULTRA_DECLARE_REFERENCE_ENTRY(SUM, MUL)                                                                     \
    int32_t ultra_rspmm_##SUM##_##MUL##_forward_cuda(                                                               \
        const int64_t *edge_index_dev, const int64_t *edge_type_dev, const void *edge_weight_dev,                   \
        const void *relation_dev, const void *input_dev, void *output_dev, int64_t num_edge, int64_t num_node,      \
        int64_t num_relation, int64_t dim, int32_t dtype, void *stream);                                            \
    int32_t ultra_rspmm_##SUM##_##MUL##_backward_cuda(                                                              \
        const int64_t *edge_index_dev, const int64_t *edge_type_dev, const void *edge_weight_dev,                   \
        const void *relation_dev, const void *input_dev, const void *output_dev, const void *output_grad_dev,       \
        void *weight_grad_dev, void *relation_grad_dev, void *input_grad_dev, int64_t num_edge, int64_t num_node,   \
        int64_t num_relation, int64_t dim, int32_t dtype, void *stream);

ULTRA_DECLARE_REFERENCE_ENTRY(add, mul) /* rspmm.h:63-68  */
ULTRA_DECLARE_REFERENCE_ENTRY(min, mul) /* rspmm.h:70-75  */
ULTRA_DECLARE_REFERENCE_ENTRY(max, mul) /* rspmm.h:77-82  */
ULTRA_DECLARE_REFERENCE_ENTRY(add, add) /* rspmm.h:84-89  */
ULTRA_DECLARE_REFERENCE_ENTRY(min, add) /* rspmm.h:91-96  */
ULTRA_DECLARE_REFERENCE_ENTRY(max, add) /* rspmm.h:98-103 */

/*
 * Static schedule of a ULTRA_PLAN_EXACT_ORDER plan when `nparts` workgroups share a span (what a launch with
 * n_span spans on a `grid`-workgroup device uses: nparts = grid / min(n_span, grid)): chain rows and group units are
 * dealt longest-processing-time-first over a cycle cost model; max_cost / mean_cost is the modelled imbalance.
 */
typedef struct {
    int32_t nparts, reserved;
    int64_t n_chunk, n_unit;                 /* chain chunks (60 edges each) and group units (4 rows each) in total */
    int64_t max_chunk_per_part, max_unit_per_part;
    double max_cost, mean_cost;              /* modelled workgroup-cycles of the fullest / the average workgroup */
} ultra_schedule_info;
/* nparts | ULTRA_SCHEDULE_12_WALKERS (both schedule calls): the schedule of the launches whose last four waves apply the layer
 * update beside the walk (ultra_rspmm_forward_update) -- the 16 streams of those waves take no rows. */
#define ULTRA_SCHEDULE_12_WALKERS (1 << 24)
int32_t ultra_plan_schedule_info(ultra_plan *plan, int32_t nparts, ultra_schedule_info *info);
/* The schedule's arrays (tests, tooling): which = 0 chunk_ptr [nparts + 1], 1 unit_ptr [nparts + 1], 2 unit ids (a unit =
 * group items n_chain_row + 4 u .. + 3 of ULTRA_ARR_ITEM), 3 chunks as {row, begin, count, flags} quadruples (flags: bit 0 first,
 * bit 1 last chunk of its row; a first chunk also holds the row's edge count in flags >> 2), 4 group-stream descriptors {first record, steps} of the
 * nparts * 64 16-lane groups, 5 stream records as (col, type) pairs: the rows of a stream back to back, each row's edges in
 * sorted order followed by a marker (row, num_relation), 6 the rows each workgroup aggregates (ascending, -1 padded to whole 32-row
 * tiles), 7 their bounds [nparts + 1].  Workgroup q of a span walks chunks [chunk_ptr[q], chunk_ptr[q + 1]),
 * then its units (C++ walk) or its 64 streams (assembly walk of the fp32 inference configuration). */
int32_t ultra_plan_schedule_export(ultra_plan *plan, int32_t nparts, int32_t which, int32_t *dst_host, int64_t capacity_elems,
                                   int64_t *count);

/* Measurement hook: while non-NULL (calling thread), the reference-order kernel stores three shader-clock samples per
 * workgroup -- start, chain rows done, end -- at trace_dev[3 * workgroup + {0, 1, 2}] (int64, device memory); further words
 * up to 32 * workgroups hold the update tail's stamps (3 * grid + 4 * workgroup + k) and each wave's end of walk
 * (8 * grid + 16 * workgroup + wave): the buffer must hold 32 * workgroups words. */
int32_t ultra_order_trace(void *trace_dev);

/* Tuning / measurement hooks. */
typedef struct {
    int32_t threads;      /* workgroup size of the main kernel (0 -> default 1024) */
    int32_t grid;         /* workgroups (0 -> one per CU) */
    int32_t rel_lds;      /* -1 auto, 0 never stage the relation slice in LDS, 1 force when it fits */
    int32_t x_lds;        /* -1 auto, 0 never stage the input slice in LDS, 1 force when it fits */
    int32_t unroll;       /* edges in flight per lane group (0 -> default) */
    int32_t reserved[3];  /* [0] != 0: ULTRA_PLAN_EXACT_ORDER plans run on the general walk kernel instead of the order kernels;
                             [1] != 0: the order kernels walk units of four rows (C++ loop) instead of group streams (assembly);
                             [2]: ultra_rspmm_forward_update -- 0 the library's choice (3 where it fits and the graph has 10+ steps a
                                  row, else 1); 1 the update in the kernel's tail; 2 beside the walk (finished rows handed over by
                                  reference and read back from memory); 3 beside the walk with the rows passing through LDS
                                  (`aggregate` is then scratch: its contents are unspecified on return); 2 / 3 where the form
                                  does not fit (LDS beside the relation slice, rows per workgroup): ULTRA_ERR_UNSUPPORTED */
} ultra_tuning;
int32_t ultra_set_tuning(const ultra_tuning *t);   /* NULL restores defaults */
int32_t ultra_get_tuning(ultra_tuning *t);

/*
 * Time the forward with HIP events on `stream` after `warmup` untimed calls.  Synchronises the stream.
 *   *ms_per_call    : mean over `iters` back-to-back calls of the whole launch sequence
 *                     (weight permute + main kernel + fix-up kernel);
 *   *ms_main_kernel : (optional) mean duration of the main rspmm_fwd_kernel alone, events recorded right
 *                     around its launch, one call at a time -- the figure bench.py's roofline block uses
 *                     and the one comparable with rocprofv3's per-kernel average.
 * point_rows_dev != NULL: `boundary` is a point boundary (see ultra_rspmm_forward_point).
 */
int32_t ultra_rspmm_forward_timed(ultra_plan *plan, int32_t sum, int32_t mul, int32_t dtype,
                                  const void *edge_weight_dev, const ultra_mat *relation, const ultra_mat *input,
                                  const ultra_mat *boundary, const int64_t *point_rows_dev, const ultra_mat *output,
                                  void *stream, int32_t warmup, int32_t iters, float *ms_per_call, float *ms_main_kernel);

/* ultra_rspmm_forward_timed for the one-launch layer (ultra_rspmm_forward_update): same two figures. */
int32_t ultra_rspmm_forward_update_timed(ultra_plan *plan, int32_t sum, int32_t mul, const ultra_mat *relation, const ultra_mat *input,
                                         const int64_t *point_rows_dev, const ultra_mat *point_values, const ultra_mat *aggregate,
                                         const void *weight, const void *bias, const void *ln_weight, const void *ln_bias, float eps,
                                         int32_t flags, const ultra_mat *output, void *stream, int32_t warmup, int32_t iters,
                                         float *ms_per_call, float *ms_main_kernel);

/* Measurement helper: streaming 16-B/lane copy of `bytes` (multiple of 16) device bytes.  Used for the
 * achievable-HBM-copy ceiling and to calibrate the FETCH_SIZE / WRITE_SIZE counters on a known byte count. */
int32_t ultra_stream_copy(void *dst_dev, const void *src_dev, int64_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ULTRA_RSPMM_H */
