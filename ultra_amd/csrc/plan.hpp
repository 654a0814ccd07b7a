// Internal definition of the aggregation plan (host side).  See include/ultra_rspmm.h.
//
// The plan replaces the per-call work of the reference: the (row, col) argsort of
// ultra/rspmm/rspmm.py:175-177 and ind2ptr of ultra/rspmm/source/rspmm.cpp:40-48.  On top of the
// CSR it carries a static, load-balanced work list for the wave64 kernel:
//
//   item  = {row, begin, len, slot}: a run of `len` sorted edges of one output row.  Rows longer
//           than seg_len are cut into several items whose partial results go to scratch slots and
//           are combined in slot order by a fix-up kernel (deterministic, no atomics).
//   unit  = the work of one wavefront: either ONE "wave item" (len > g_max; the four 16-lane
//           groups stride its edges and reduce across groups at the end) or FOUR "group items"
//           (len <= g_max; each 16-lane group walks one item sequentially, in sorted order).
//   Items of each class are ordered by descending length so that the four items of a unit have
//   similar trip counts and the cyclic unit -> wave assignment is longest-first.
#pragma once

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ultra_rspmm.h"

namespace ultra {

struct Item {
    int32_t row, begin, len, slot;
};

// Reference-order plans (ULTRA_PLAN_EXACT_ORDER): rows with more than chain_min edges are "chain" rows.  A workgroup
// walks one in chunks of CHAIN_SLOTS edges: fifteen producer waves (sixty 16-lane groups) compute one message each and
// park it in an LDS ring, one consumer wave adds the parked messages in sorted edge order -- the reference's
// sequential summation order (rspmm.cpp:61-72) without a 9,000-step walk by a single lane group.
struct Chunk {
    int32_t row, begin, count, flags;   // flags: CHUNK_FIRST / CHUNK_LAST chunk of its row; a first chunk: | row length << 2
};
enum { CHUNK_FIRST = 1, CHUNK_LAST = 2, CHUNK_LEN_SHIFT = 2 };
// ULTRA_ORDER_WAVES: waves per workgroup of the reference-order kernels (16 = 1024 threads: a workgroup owns its CU).  12
// (768 threads, 44-edge chunks) leaves a CU four wave slots, a quarter of its registers and ~20 KB of LDS: room for the
// latency-bound kernels of ANOTHER batch's forward (the relation-graph layers) to run beside the walk -- see DESIGN.md 3.9.
#ifndef ULTRA_ORDER_WAVES
#define ULTRA_ORDER_WAVES 16
#endif
constexpr int ORDER_WAVES = ULTRA_ORDER_WAVES;
constexpr int CHAIN_SLOTS = 4 * (ORDER_WAVES - 1);   // producer groups of a workgroup (every wave but the consumer, x 4)
constexpr int ORDER_GROUPS = 4 * ORDER_WAVES;        // 16-lane groups of a workgroup
// (Round 3 built the chain and stream phases SIDE BY SIDE -- wave 0 consuming, eight producer waves, seven walking from the start,
// LDS words instead of the barrier -- as a measurement build: bit-exact, the average workgroup 8 % earlier, but the chain itself
// 675 -> 800 cycles per chunk and the 9,067-edge row's workgroup the launch: - 1.4 % on the step.  The build was removed in round
// 5; its numbers are in DESIGN.md 8 and profiles/r3_*.)
constexpr int CHUNK_PAD = 32;           // descriptors readable behind a schedule's last chunk (>= 3 x the producers' depth)
// Record format of the group streams of the TWELVE-walker schedules (the update-beside-the-walk launches; nothing else reads
// them): 1 = (col * 256, type * 256) -- these launches serve whole-span rows of 256 bytes only, so gather offset and relation-row
// address are one DPP-fused add each in the generated walk (tools/gen_order_asm.py, DIET) -- 0 = (col, type) like the
// sixteen-walker schedules, whose walks serve any row stride.  Both sides of 2^24 rows / relations are excluded by the caller.
#ifndef ULTRA_STREAM_PRESHIFT
#define ULTRA_STREAM_PRESHIFT 1
#endif
constexpr int ORDER_PAD = 128;           // the device record / perm streams are readable this many entries past the last edge

// Static work assignment of one launch geometry: `nparts` workgroups share the items of a span.  Built on first use
// (longest-processing-time-first over a cost model of chain rows and group units), then kept on the device.
struct Schedule {
    int32_t nparts = 0;
    std::vector<int32_t> chunk_ptr, unit_ptr, units;   // [nparts + 1], [nparts + 1], unit ids in launch order
    std::vector<Chunk> chunks;
    // Group STREAMS (the assembly walk of the inference configuration): each of the ORDER_GROUPS 16-lane groups of a
    // workgroup gets a sequence of whole rows; `srec` holds, stream after stream, the rows' (col, type) records in sorted
    // edge order, each row closed by a marker record (row, num_rel); sdesc[part * ORDER_GROUPS + group] = {first record, steps}.
    std::vector<int32_t> srec, sdesc;
    // Rows each workgroup aggregates (its chain rows and the rows of its streams), ascending, padded with -1 to whole
    // 32-row tiles: prow[prow_ptr[part] .. prow_ptr[part + 1]) -- the work list of the update the workgroup applies to its
    // own rows after the walk (rspmm_order_kernels.hpp, UPDATE).
    std::vector<int32_t> prow, prow_ptr;
    int32_t *d_chunk_ptr = nullptr, *d_unit_ptr = nullptr, *d_units = nullptr, *d_srec = nullptr, *d_sdesc = nullptr;
    int32_t *d_prow = nullptr, *d_prow_ptr = nullptr;
    Chunk *d_chunks = nullptr;
    double max_cost = 0.0, mean_cost = 0.0;             // cost model's load of the fullest / average workgroup
    int32_t max_rows = 0;                               // most rows any one workgroup aggregates
    int32_t max_chain_rows = 0;                         // most chain rows any one workgroup walks
    int32_t max_stream_steps = 0;                       // longest group stream in steps (edges + markers)
    int32_t rec_shift = 0;                              // srec holds (col << rec_shift, type << rec_shift): ULTRA_STREAM_PRESHIFT
    int64_t srec_pad = 2 * ORDER_PAD;                   // readable zeros (int32 words) behind the last stream's records
};

struct DevicePlan {
    int32_t *row_ptr = nullptr, *col = nullptr, *type = nullptr, *perm = nullptr, *erow = nullptr;
    int32_t *rec = nullptr;      // (col, type) pairs per sorted edge: the record stream of the reference-order kernel
    uint32_t *packed = nullptr;
    Item *items = nullptr;
    int32_t *split_row = nullptr, *split_ptr = nullptr;
    uint8_t *a_frag = nullptr;   // ULTRA_PLAN_DENSE
    uint8_t *a16 = nullptr;         // ULTRA_PLAN_DENSE, 16-row tiles (fused layer kernel)
    uint8_t *a_ex = nullptr;        // ULTRA_PLAN_DENSE, reference-order layer kernel (dense_order_layer.hip)
    uint8_t *self_loop = nullptr;   // per node: bit 0 = has an edge onto itself, bit 1 = has an in-edge from another node (layer-0 path)
    void *w_sorted = nullptr;
    size_t w_sorted_bytes = 0;
    void *w_sorted_cap = nullptr;     // the permuted weights of launches recorded into a hipGraph (replays never touch w_sorted)
    size_t w_sorted_cap_bytes = 0;
    void *partial = nullptr;
    size_t partial_bytes = 0;
    // backward of the rspmm on listed rows as gathers (rows_bwd_kernels.hpp): the edge list grouped by source / by type, cut into
    // segments; built on first use
    void *rb_rec_c = nullptr, *rb_seg_c = nullptr, *rb_multi_c = nullptr;
    void *rb_rec_t = nullptr, *rb_seg_t = nullptr, *rb_multi_t = nullptr;
    void *rb_work = nullptr;
    size_t rb_work_bytes = 0;
    int device = -1;
};

void set_error(const std::string &msg);

// ULTRA_PLAN_DENSE: the number of 8-column groups is padded (with empty cells) to a multiple of this, so that each of
// the four waves of a workgroup runs whole software-pipeline blocks (rspmm_dense.hip)
#define ULTRA_DENSE_KG_ALIGN 20

}  // namespace ultra

struct ultra_plan {
    int64_t num_edge = 0, num_out = 0, num_in = 0, num_rel = 0;
    int32_t seg_len = 256, g_max = 64, flags = 0;
    int32_t type_bits = 0;
    bool packed_ok = false;
    // the weight vector whose permutation d.w_sorted holds (ultra_rspmm_weight_epoch): its tag, address, dtype, stream
    int64_t w_epoch = 0;
    const void *w_src = nullptr;
    int32_t w_dtype = -1;
    void *w_stream = nullptr;
    // ... and the same for d.w_sorted_cap, valid inside the stream capture w_cap_id only
    int64_t w_cap_epoch = 0;
    const void *w_cap_src = nullptr;
    int32_t w_cap_dtype = -1;
    void *w_cap_stream = nullptr;
    unsigned long long w_cap_id = 0;
    int32_t max_row_len = -1;     // longest row (edges); computed on first use (the layer-0 launch sizes its grid with it)
    // rows-backward index (d.rb_*): segments / owners with several segments / partial rows, by source and by type
    bool rb_built = false;
    int64_t rb_n_seg_c = 0, rb_n_multi_c = 0, rb_n_part_c = 0, rb_n_seg_t = 0, rb_n_multi_t = 0, rb_n_part_t = 0;

    std::vector<int32_t> row_ptr, col, type, perm, erow;  // erow: output row of each sorted edge
    std::vector<uint32_t> packed;
    std::vector<ultra::Item> items;
    int64_t n_w = 0, n_g = 0, n_unit = 0;
    // ULTRA_PLAN_EXACT_ORDER: items[0, n_chain) are the chain rows (longest first), group items follow; `rec` interleaves
    // (col, type); schedules are keyed by nparts
    int64_t n_chain = 0;
    int32_t chain_min = 256;
    std::vector<int32_t> rec;
    std::map<int32_t, ultra::Schedule *> schedules;
    std::mutex sched_mu;
    std::vector<int32_t> split_row, split_ptr;
    int64_t n_slot = 0;
    int64_t n_type_run = 0;

    // ULTRA_PLAN_DENSE: edge multiplicities as bytes, [row_tile][type_chunk][kgroup][lane][type_in_chunk][q]
    // (see ULTRA_ARR_DENSE); dense_tc = types per chunk (1, 2 or 4), dense_ntc = number of chunks
    std::vector<uint8_t> a_frag;
    int32_t dense_rt = 0, dense_kg = 0, dense_tc = 0, dense_ntc = 0;
    bool dense_overflow = false;   // some multiplicity exceeds 255
    // the same multiplicities for 16-row tiles and v_mfma_f32_16x16x4_f32 (dense_layer.hip), at most 4 relation types:
    // [row_tile16][chunk of 16 columns][lane = (row % 16) + 16 (col % 4)][step = (col % 16) / 4][type]; a16_chunks % 8 == 0
    std::vector<uint8_t> a16;
    int32_t a16_chunks = 0;
    // the adjacency for the reference-order layer kernel (dense_order_layer.hip): [row_tile16][chunk of 16 columns]
    // [lane = (row % 16) + 16 type][col % 16], one byte (0 / 1) per cell.  Only built when every (row, col) pair lists its
    // parallel edges in ascending type order without repeats (then the kernel's k-ordered fma chain IS the reference's
    // summation order); empty otherwise.
    std::vector<uint8_t> a_ex;

    std::vector<uint8_t> self_loop;   // [num_out] built with the edge list (square graphs): bit 0 self loop, bit 1 in-edge from another node

    // original (unsorted) edges, kept to derive the backward plans lazily
    std::vector<int32_t> h_row, h_col, h_type;

    bool on_device = false;
    int32_t pinned = 0;     // > 0: a captured hipGraph holds this plan's device pointers -- scratch buffers must not move
    ultra::DevicePlan d;

    // backward plans (built on first use):
    //   tplan: rows = edge_index[1], sources = edge_index[0]  -> input_grad
    //   rplan: rows = edge_type, "relation" index = edge_index[0], sources = edge_index[1] -> relation_grad
    ultra_plan *tplan = nullptr;
    ultra_plan *rplan = nullptr;
};

namespace ultra {

// Build a plan from int32 (row, col, type) triples.  Pure host code.
ultra_plan *build_plan(const int32_t *row, const int32_t *col, const int32_t *type, int64_t num_edge,
                       int64_t num_out, int64_t num_in, int64_t num_rel, const ultra_plan_opts *opts,
                       bool keep_edges);

// The schedule of a reference-order plan for `nparts` workgroups per span (host arrays only; cached in the plan by
// the caller under sched_mu).
// walkers: 16 = every wave of a workgroup walks streams; 12 = the last four waves take no rows (they apply the layer update
// beside the walk).
Schedule *build_schedule(const ultra_plan *p, int32_t nparts, int32_t walkers = 16);

}  // namespace ultra
