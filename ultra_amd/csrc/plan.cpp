// Host-side plan builder: CSR + load-balanced work list.  No HIP calls in this file.
// Replaces ultra/rspmm/rspmm.py:175-177 (per-call argsort) and rspmm.cpp:40-48 (per-call ind2ptr).
#include "plan.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <queue>

namespace ultra {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

static int bits_for(int64_t n) {  // bits needed to store values in [0, n)
    int b = 0;
    while ((int64_t(1) << b) < n) ++b;
    return b;
}

// Stable counting sort of `order` by key[order[i]] in [0, nkey).
static void counting_pass(std::vector<int32_t> &order, const int32_t *key, int64_t nkey) {
    const int64_t n = (int64_t)order.size();
    std::vector<int64_t> start((size_t)nkey + 1, 0);
    for (int64_t i = 0; i < n; ++i) start[(size_t)key[order[i]] + 1]++;
    for (int64_t k = 0; k < nkey; ++k) start[k + 1] += start[k];
    std::vector<int32_t> out((size_t)n);
    for (int64_t i = 0; i < n; ++i) out[(size_t)start[key[order[i]]]++] = order[i];
    order.swap(out);
}

ultra_plan *build_plan(const int32_t *row, const int32_t *col, const int32_t *type, int64_t E, int64_t num_out,
                       int64_t num_in, int64_t num_rel, const ultra_plan_opts *opts, bool keep_edges) {
    ultra_plan *p = new ultra_plan();
    p->num_edge = E;
    p->num_out = num_out;
    p->num_in = num_in;
    p->num_rel = num_rel;
    if (opts) {
        if (opts->seg_len > 0) p->seg_len = opts->seg_len;
        if (opts->g_max > 0) p->g_max = opts->g_max;
        p->flags = opts->flags;
    }
    const bool exact = (p->flags & ULTRA_PLAN_EXACT_ORDER) != 0;
    const bool type_runs = !exact && (p->flags & ULTRA_PLAN_TYPE_RUNS) != 0;
    if (p->g_max > p->seg_len) p->g_max = p->seg_len;
    if (exact) p->chain_min = p->seg_len;   // rows longer than this are walked by a whole workgroup (plan.hpp)

    if (p->flags & ULTRA_PLAN_DENSE) {
        // dense format: only the multiplicity matrices, as bytes, laid out the way the MFMA A-operand is consumed
        // (lane l of a 32x32x2 step holds A[i = l % 32][k = l / 32]; one load of 4 * tc bytes per lane covers the
        // four steps of an 8-column group for tc relation types)
        p->dense_rt = (int32_t)((num_out + 31) / 32);
        p->dense_kg = (int32_t)(((num_in + 7) / 8 + ULTRA_DENSE_KG_ALIGN - 1) / ULTRA_DENSE_KG_ALIGN * ULTRA_DENSE_KG_ALIGN);
        p->dense_tc = num_rel <= 1 ? 1 : (num_rel == 2 ? 2 : 4);
        p->dense_ntc = (int32_t)std::max<int64_t>(1, (num_rel + p->dense_tc - 1) / p->dense_tc);
        const int64_t tc = p->dense_tc;
        p->a_frag.assign((size_t)p->dense_rt * p->dense_ntc * p->dense_kg * 64 * 4 * tc, 0);
        for (int64_t e = 0; e < E; ++e) {
            const int64_t rt = row[e] / 32, i = row[e] % 32, kg = col[e] / 8, within = col[e] % 8;
            const int64_t lane = (within % 2) * 32 + i, q = within / 2, chunk = type[e] / tc, tl = type[e] % tc;
            uint8_t &cell = p->a_frag[(size_t)(((((rt * p->dense_ntc + chunk) * p->dense_kg + kg) * 64 + lane) * tc + tl) * 4 + q)];
            if (cell == 255) p->dense_overflow = true; else ++cell;
        }
        if (num_rel <= 4 && num_out == num_in) {
            const int64_t rt16 = (num_out + 15) / 16;
            p->a16_chunks = (int32_t)(((num_in + 15) / 16 + 7) / 8 * 8);   // whole 2-chunk stages per k-quarter
            p->a16.assign((size_t)rt16 * p->a16_chunks * 64 * 16, 0);
            for (int64_t e = 0; e < E; ++e) {
                const int64_t rt = row[e] / 16, i = row[e] % 16, chunk = col[e] / 16, within = col[e] % 16;
                const int64_t lane = i + 16 * (within % 4), step = within / 4;
                uint8_t &cell = p->a16[(size_t)((((rt * p->a16_chunks + chunk) * 64 + lane) * 4 + step) * 4 + type[e])];
                if (cell < 255) ++cell;   // (overflow already flagged above)
            }
        }
        if (num_rel <= 4 && num_out == num_in && num_out * num_in <= (int64_t(1) << 26)) {
            // reference-order layer kernel: parallel edges of a (row, col) pair must come in ascending type order
            // (original edge order = the stable sort's order), each type at most once
            std::vector<int8_t> last((size_t)(num_out * num_in), (int8_t)-1);
            bool ok = true;
            for (int64_t e = 0; e < E && ok; ++e) {
                int8_t &l = last[(size_t)(row[e] * num_in + col[e])];
                if (type[e] <= l) ok = false;
                l = (int8_t)type[e];
            }
            if (ok) {
                const int64_t rt16 = (num_out + 15) / 16, njc = (num_in + 15) / 16;
                p->a_ex.assign((size_t)(rt16 * njc * 64 * 16), 0);
                for (int64_t e = 0; e < E; ++e) {
                    const int64_t rt = row[e] / 16, i = row[e] % 16, jc = col[e] / 16, q = col[e] % 16;
                    p->a_ex[(size_t)((((rt * njc + jc) * 64) + i + 16 * type[e]) * 16 + q)] = 1;
                }
            }
        }
        p->split_ptr.push_back(0);
        return p;
    }

    // ---- sort by (row, [type,] col), stable in the original edge id: LSD counting passes ----
    std::vector<int32_t> order((size_t)E);
    std::iota(order.begin(), order.end(), 0);
    if (E > 0) {
        counting_pass(order, col, num_in);
        if (type_runs) counting_pass(order, type, num_rel);
        counting_pass(order, row, num_out);
    }
    p->perm = order;
    p->col.resize((size_t)E);
    p->type.resize((size_t)E);
    p->erow.resize((size_t)E);
    p->row_ptr.assign((size_t)num_out + 1, 0);
    for (int64_t k = 0; k < E; ++k) {
        const int32_t e = order[(size_t)k];
        p->col[(size_t)k] = col[e];
        p->type[(size_t)k] = type[e];
        p->erow[(size_t)k] = row[e];
        p->row_ptr[(size_t)row[e] + 1]++;
    }
    for (int64_t r = 0; r < num_out; ++r) p->row_ptr[(size_t)r + 1] += p->row_ptr[(size_t)r];

    // ---- packed (col, type) word when both fit into 32 bits ----
    p->type_bits = bits_for(std::max<int64_t>(num_rel, 1));
    const int col_bits = bits_for(std::max<int64_t>(num_in, 1));
    p->packed_ok = (p->type_bits + col_bits) <= 32;
    if (p->packed_ok) {
        p->packed.resize((size_t)E);
        for (int64_t k = 0; k < E; ++k)
            p->packed[(size_t)k] = ((uint32_t)p->col[(size_t)k] << p->type_bits) | (uint32_t)p->type[(size_t)k];
    }

    // ---- number of (row, type) runs (a plan statistic: long runs favour ULTRA_PLAN_TYPE_RUNS) ----
    {
        std::vector<int32_t> last_row((size_t)std::max<int64_t>(num_rel, 1), -1);
        int64_t runs = 0;
        for (int64_t k = 0; k < E; ++k) {
            const int32_t t = p->type[(size_t)k], r = p->erow[(size_t)k];
            if (last_row[(size_t)t] != r) {
                last_row[(size_t)t] = r;
                ++runs;
            }
        }
        p->n_type_run = runs;
    }

    if (exact) {
        p->rec.resize((size_t)E * 2);
        for (int64_t k = 0; k < E; ++k) {
            p->rec[(size_t)2 * k] = p->col[(size_t)k];
            p->rec[(size_t)2 * k + 1] = p->type[(size_t)k];
        }
    }

    // ---- items ----
    std::vector<Item> witems, gitems, chain;
    int32_t next_slot = 0;
    p->split_ptr.push_back(0);
    for (int64_t r = 0; r < num_out; ++r) {
        const int32_t b = p->row_ptr[(size_t)r], e = p->row_ptr[(size_t)r + 1];
        const int32_t deg = e - b;
        if (type_runs) {
            // one or more items per (row, type) run; a row with a single item writes its output directly
            std::vector<Item> mine;
            int32_t pos = b;
            while (pos < e) {
                int32_t end = pos + 1;
                while (end < e && p->type[(size_t)end] == p->type[(size_t)pos]) ++end;
                const int32_t len_run = end - pos;
                const int32_t nseg = (len_run + p->seg_len - 1) / p->seg_len;
                const int32_t base = len_run / nseg, extra = len_run % nseg;
                for (int32_t s = 0; s < nseg; ++s) {
                    const int32_t len = base + (s < extra ? 1 : 0);
                    mine.push_back(Item{(int32_t)r, pos, len, -1});
                    pos += len;
                }
            }
            if (mine.empty()) mine.push_back(Item{(int32_t)r, b, 0, -1});
            if (mine.size() > 1) {
                for (auto &it : mine) it.slot = next_slot++;
                p->split_row.push_back((int32_t)r);
                p->split_ptr.push_back(next_slot);
            }
            for (auto &it : mine) (it.len <= p->g_max ? gitems : witems).push_back(it);
            continue;
        }
        if (exact) {
            // one item per row, never split: the walk order is the sorted edge order.  Long rows are chain rows.
            Item it{(int32_t)r, b, deg, -1};
            (deg > p->chain_min ? chain : gitems).push_back(it);
        } else if (deg <= p->seg_len) {
            Item it{(int32_t)r, b, deg, -1};
            if (deg <= p->g_max)
                gitems.push_back(it);
            else
                witems.push_back(it);
        } else {
            // balanced segments of at most seg_len edges
            const int32_t nseg = (deg + p->seg_len - 1) / p->seg_len;
            const int32_t base = deg / nseg, extra = deg % nseg;
            int32_t pos = b;
            for (int32_t s = 0; s < nseg; ++s) {
                const int32_t len = base + (s < extra ? 1 : 0);
                Item it{(int32_t)r, pos, len, next_slot++};
                pos += len;
                if (len <= p->g_max)
                    gitems.push_back(it);
                else
                    witems.push_back(it);
            }
            p->split_row.push_back((int32_t)r);
            p->split_ptr.push_back(next_slot);
        }
    }
    p->n_slot = next_slot;
    auto by_len_desc = [](const Item &a, const Item &b) { return a.len > b.len; };
    std::stable_sort(witems.begin(), witems.end(), by_len_desc);
    std::stable_sort(gitems.begin(), gitems.end(), by_len_desc);
    std::stable_sort(chain.begin(), chain.end(), by_len_desc);
    // (kernels without the chain walk see the chain rows as ordinary -- very long -- group items at the head of the list)
    p->n_chain = (int64_t)chain.size();
    p->n_w = (int64_t)witems.size();
    p->n_g = (int64_t)(chain.size() + gitems.size());
    p->n_unit = p->n_w + (p->n_g + 3) / 4;
    p->items.reserve(chain.size() + witems.size() + gitems.size());
    p->items.insert(p->items.end(), chain.begin(), chain.end());
    p->items.insert(p->items.end(), witems.begin(), witems.end());
    p->items.insert(p->items.end(), gitems.begin(), gitems.end());

    if (keep_edges) {
        p->self_loop.assign((size_t)std::max<int64_t>(num_out, 1), 0);
        for (int64_t e = 0; e < E; ++e)
            if (row[e] < num_out) p->self_loop[(size_t)row[e]] |= (row[e] == col[e]) ? 1 : 2;   // bit 0: onto itself, bit 1: from another node
        p->h_row.assign(row, row + E);
        p->h_col.assign(col, col + E);
        p->h_type.assign(type, type + E);
    }
    return p;
}

// ---- schedule of a reference-order plan ----
// Cost model in workgroup-cycles (calibrated on MI355X): a chain row costs its consumer wave one
// dependent add per edge plus a ring hand-over per chunk; a group unit costs one wave (a sixteenth of the workgroup's
// issue slots) one walk step per edge of its longest row.
static double COST_CHAIN_EDGE = 9.0, COST_CHAIN_CHUNK = 250.0, COST_CHAIN_ROW = 2000.0;
static double COST_UNIT_STEP = 28.0, COST_UNIT = 100.0;
// the stream walk (assembly paths; order_trace on MI355X): workgroup cycles per chain chunk / chain row, and per step of
// one group stream when all sixteen waves walk (21.5 cycles per wave step / 4 groups)
static double COST_S_CHUNK = 675.0, COST_S_ROW = 1380.0, COST_S_STEP = 6.0;
// Share of a workgroup's stream work per wave quartet (waves 0-3, 4-7, 8-11, 12-15; mean 1).  The CU issues oldest wave
// first: with equal shares the four quartets finish their walks at 48 k / 59 k / 76 k / 89 k cycles (tools/overlap_trace.py,
// FB15k237 bs 8) and the last one walks alone at a quarter of the CU's gather rate.  Rates by age rank solved from those
// four times come out as 1 : 0.77 : 0.47 : 0.35, but the young waves are nearly starved while older ones run, so the shares were
// swept instead (tools/share_sweep.sh): 1.7 / 1.3 / 0.7 / 0.3 lets the quartets finish at 84 k / 80 k / 82 k / 85 k -- the
// launch's last workgroup at 142.5 k cycles instead of 147.7 k, 69.5 us instead of 71.8 stand-alone.  (s_setprio against
// the age order only reverses who finishes first: 50 k / 60 k / 76 k / 89 k from the youngest quartet up.)
static double WAVE_SHARE[4] = {1.7, 1.3, 0.7, 0.3};
// ... and when the youngest quartet does not walk at all (its waves apply the layer update beside the walk:
// rspmm_order_kernel, UPDATE == 2): the same age effect among the three walking quartets
static double WAVE_SHARE_12[4] = {1.35, 1.15, 0.9, 0.0};
// ... or per walker wave (calibration runs: ULTRA_STREAM_SHARES_WAVES_12="w0,...,w11"; negative = unset: the quartet's share)
static double WAVE_SHARE_12_WAVES[12] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
// A row the plan lists as a chain row (> chain_min = 256 edges) is still better off as ONE group stream's row when the streams
// of the launch are long enough to take it: the chain sums an edge in 11 workgroup-cycles, a stream in 6-8.  In the schedules of
// the update-beside-the-walk launches (twelve walkers; nothing else reads them) chain rows of up to CHAIN_LIMIT_FACTOR x the mean
// stream length go to the streams.  tools/step_probe.py PROBE_SEG_LEN (FB15k237 shape, ms per step, one batch at a time with 32
// partitions a sample: mean stream 364 steps / two in flight with 24: 485 steps): threshold 256: 0.667 / 0.605, 384: 0.661 /
// 0.596, 512: 0.655 / 0.595, 768: 0.648 / 0.590, 1,024: 0.705 / 0.580, 1,536: 0.837 / 0.597 -- the best is at 2.1 x in both.
static double CHAIN_LIMIT_FACTOR = 2.1;
// What a ROW costs a stream of the twelve-walker schedules beyond its edges, in steps, per wave quartet.  A flush is not a step
// like the others: the wave parks the row in the hand-off ring -- one LDS round trip, sometimes a wait for a free ring row --
// and the stall holds up all four streams of the wave, whatever the wave's age; the steps themselves go faster the older the
// wave is (WAVE_SHARE_12).  Fitted from a traced launch (tools/wave_fit.py: walk = c_q steps + r rows per wave; a stream's share
// of its wave's row stalls is 4 r / c_q steps a row).  ULTRA_STREAM_ROW_COST_12="r0,r1,r2" overrides (calibration runs).
static double ROW_COST_12[4] = {10.0, 5.0, 0.0, 0.0};
// (Measured and dropped in round 5: dealing the rows to the PARTITIONS first -- by steps + a row cost against the partition's budget
// -- and then to each partition's streams: a long row then lands in a partition whose streams are shorter than it is (one stream
// of 687 steps beside 500: 86 -> 95 us per layer at the FB15k237 shape); the one-level deal below places the long rows first,
// across all streams of the launch.  profiles/r5_experiments.txt.)

static void read_cost_override() {   // calibration runs only: ULTRA_SCHED_COSTS="edge,chunk,row,step,unit", ULTRA_STREAM_COSTS="chunk,row,step"
    const char *env = std::getenv("ULTRA_SCHED_COSTS");
    double v[5];
    if (env && std::sscanf(env, "%lf,%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5) {
        COST_CHAIN_EDGE = v[0], COST_CHAIN_CHUNK = v[1], COST_CHAIN_ROW = v[2], COST_UNIT_STEP = v[3], COST_UNIT = v[4];
    }
    env = std::getenv("ULTRA_STREAM_SHARES");   // calibration runs: "q0,q1,q2,q3"
    if (env && std::sscanf(env, "%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3]) == 4) {
        const double mean = (v[0] + v[1] + v[2] + v[3]) / 4.0;
        for (int k = 0; k < 4; ++k) WAVE_SHARE[k] = mean > 0 ? v[k] / mean : 1.0;
    }
    env = std::getenv("ULTRA_STREAM_SHARES_12");   // calibration runs: "q0,q1,q2" (the update waves take no rows)
    if (env && std::sscanf(env, "%lf,%lf,%lf", &v[0], &v[1], &v[2]) == 3) {
        for (int k = 0; k < 3; ++k) WAVE_SHARE_12[k] = v[k];
        WAVE_SHARE_12[3] = 0.0;
    }
    env = std::getenv("ULTRA_STREAM_SHARES_WAVES_12");
    if (env) {
        double w[12];
        if (std::sscanf(env, "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", &w[0], &w[1], &w[2], &w[3], &w[4], &w[5], &w[6], &w[7], &w[8],
                        &w[9], &w[10], &w[11]) == 12)
            for (int k = 0; k < 12; ++k) WAVE_SHARE_12_WAVES[k] = w[k];
    }
    env = std::getenv("ULTRA_STREAM_ROW_COST_12");
    if (env && std::sscanf(env, "%lf,%lf,%lf", &v[0], &v[1], &v[2]) == 3) {
        for (int k = 0; k < 3; ++k) ROW_COST_12[k] = v[k];
    }
    env = std::getenv("ULTRA_CHAIN_LIMIT_FACTOR");   // calibration runs (0: every listed chain row stays one)
    if (env && std::sscanf(env, "%lf", &v[0]) == 1) CHAIN_LIMIT_FACTOR = v[0];
    env = std::getenv("ULTRA_STREAM_COSTS");
    if (env) {
        const int n = std::sscanf(env, "%lf,%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3], &v[4]);
        if (n >= 3) COST_S_CHUNK = v[0], COST_S_ROW = v[1], COST_S_STEP = v[2];
    }
}

Schedule *build_schedule(const ultra_plan *p, int32_t nparts, int32_t walkers) {
    read_cost_override();
    Schedule *s = new Schedule();
    s->nparts = nparts;
    const double *wave_share = walkers == 12 ? WAVE_SHARE_12 : WAVE_SHARE;
    const int64_t n_chain = p->n_chain, n_item = (int64_t)p->items.size();
    const int64_t n_unit = (n_item - n_chain + 3) / 4;
    struct Work {
        double cost;
        int32_t id;   // chain item index, or n_chain + unit id
    };
    // chain rows that this launch size's streams take as plain rows (twelve-walker schedules only: the sixteen-walker ones also
    // serve the unit walk, whose units do not list them)
    std::vector<char> to_stream((size_t)n_chain, 0);
    if (walkers == 12 && CHAIN_LIMIT_FACTOR > 0.0) {
        double all_steps = 0.0;
        for (int64_t g = 0; g < n_item; ++g) all_steps += p->items[(size_t)g].len + 1;
        const double limit = CHAIN_LIMIT_FACTOR * all_steps / ((double)nparts * 16.0 * (walkers / 4));   // x the mean stream length
        for (int64_t c = 0; c < n_chain; ++c) to_stream[(size_t)c] = p->items[(size_t)c].len <= limit;
    }
    std::vector<Work> work;
    work.reserve((size_t)(n_chain + n_unit));
    for (int64_t c = 0; c < n_chain; ++c) {
        if (to_stream[(size_t)c]) continue;
        const int32_t len = p->items[(size_t)c].len;
        const int32_t nchunk = (len + CHAIN_SLOTS - 1) / CHAIN_SLOTS;
        work.push_back(Work{COST_CHAIN_ROW + COST_CHAIN_EDGE * len + COST_CHAIN_CHUNK * nchunk, (int32_t)c});
    }
    for (int64_t u = 0; u < n_unit; ++u) {
        const int32_t steps = p->items[(size_t)(n_chain + 4 * u)].len;   // group items are sorted by descending length
        work.push_back(Work{COST_UNIT + COST_UNIT_STEP * steps, (int32_t)(n_chain + u)});
    }
    std::stable_sort(work.begin(), work.end(), [](const Work &a, const Work &b) { return a.cost > b.cost; });
    // longest processing time first onto the least loaded workgroup
    typedef std::pair<double, int32_t> Load;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int32_t q = 0; q < nparts; ++q) heap.push(Load(0.0, q));
    std::vector<std::vector<int32_t>> part_chain((size_t)nparts), part_unit((size_t)nparts);
    std::vector<double> load((size_t)nparts, 0.0);
    for (const Work &w : work) {
        Load l = heap.top();
        heap.pop();
        (w.id < n_chain ? part_chain : part_unit)[(size_t)l.second].push_back(w.id < n_chain ? w.id : w.id - (int32_t)n_chain);
        l.first += w.cost;
        load[(size_t)l.second] = l.first;
        heap.push(l);
    }
    s->chunk_ptr.assign((size_t)nparts + 1, 0);
    s->unit_ptr.assign((size_t)nparts + 1, 0);
    for (int32_t q = 0; q < nparts; ++q) {
        for (int32_t c : part_chain[(size_t)q]) {
            const Item &it = p->items[(size_t)c];
            for (int32_t pos = 0; pos < it.len; pos += CHAIN_SLOTS) {
                const int32_t cnt = std::min<int32_t>(CHAIN_SLOTS, it.len - pos);
                // (a row's first chunk also carries the row's length: the consumer wave reads one descriptor per row)
                s->chunks.push_back(Chunk{it.row, it.begin + pos, cnt,
                                          (pos == 0 ? (CHUNK_FIRST | (it.len << CHUNK_LEN_SHIFT)) : 0) |
                                              (pos + cnt == it.len ? CHUNK_LAST : 0)});
            }
        }
        s->chunk_ptr[(size_t)q + 1] = (int32_t)s->chunks.size();
        s->units.insert(s->units.end(), part_unit[(size_t)q].begin(), part_unit[(size_t)q].end());
        s->unit_ptr[(size_t)q + 1] = (int32_t)s->units.size();
        s->max_cost = std::max(s->max_cost, load[(size_t)q]);
        s->mean_cost += load[(size_t)q] / nparts;
    }
    // ---- group streams ----
    // Workgroup q may spend T - (its chain work) on its streams, T = the mean total; its 64 streams share that budget
    // equally.  Rows go longest first onto the stream that stays relatively emptiest ((load + len) / budget).
    {
        std::vector<double> chain_cost((size_t)nparts, 0.0);
        double total = 0.0;
        for (int32_t q = 0; q < nparts; ++q) {
            for (int32_t c : part_chain[(size_t)q]) {
                const int32_t len = p->items[(size_t)c].len;
                chain_cost[(size_t)q] += COST_S_ROW + COST_S_CHUNK * ((len + CHAIN_SLOTS - 1) / CHAIN_SLOTS);
            }
            total += chain_cost[(size_t)q];
        }
        double steps = 0.0;
        for (int64_t g = 0; g < n_item; ++g)
            if (g >= n_chain || to_stream[(size_t)g]) steps += p->items[(size_t)g].len + 1;
        const int64_t nstream = (int64_t)nparts * ORDER_GROUPS;
        std::vector<double> weight((size_t)nstream);
        total += COST_S_STEP * steps;
        const double T = total / nparts;
        for (int32_t q = 0; q < nparts; ++q) {
            const double budget = std::max(T - chain_cost[(size_t)q], 0.02 * T);   // (never zero: every row needs a home)
            for (int g = 0; g < ORDER_GROUPS; ++g) {
                double share = wave_share[g / 16];
                if (walkers == 12 && g / 4 < 12 && WAVE_SHARE_12_WAVES[g / 4] >= 0.0) share = WAVE_SHARE_12_WAVES[g / 4];
                weight[(size_t)q * ORDER_GROUPS + g] = budget * share;
            }
        }
        typedef std::pair<double, int64_t> Slot;   // ((load + 1) / weight, stream): the heap's top is the relatively emptiest
        std::priority_queue<Slot, std::vector<Slot>, std::greater<Slot>> sheap;
        std::vector<int64_t> sload((size_t)nstream, 0);
        std::vector<double> scost((size_t)nstream, 0.0);   // steps + what the rows cost beyond them (ROW_COST_12)
        std::vector<std::vector<int32_t>> srows((size_t)nstream);
        const auto deal_into = [&](const int64_t stream, const int64_t g) {
            srows[(size_t)stream].push_back((int32_t)g);
            sload[(size_t)stream] += p->items[(size_t)g].len + 1;
            scost[(size_t)stream] += p->items[(size_t)g].len + 1 + (walkers == 12 ? ROW_COST_12[(stream % ORDER_GROUPS) / 16] : 0.0);
        };
        {
        for (int64_t g = 0; g < nstream; ++g)
            if (weight[(size_t)g] > 0.0) sheap.push(Slot(1.0 / weight[(size_t)g], g));   // (a stream without a share takes no rows)
        for (int64_t g = 0; g < n_item; ++g) {   // (chain items that go to the streams first: longest first throughout)
            if (g < n_chain && !to_stream[(size_t)g]) continue;
            const Slot sl = sheap.top();
            sheap.pop();
            deal_into(sl.second, g);
            sheap.push(Slot((scost[(size_t)sl.second] + 1.0) / weight[(size_t)sl.second], sl.second));
        }
        }
        s->sdesc.assign((size_t)nstream * 2, 0);
        s->srec.reserve((size_t)(2 * (int64_t)steps) + 2 * ORDER_PAD);
        for (int64_t g = 0; g < nstream; ++g) {
            s->sdesc[(size_t)2 * g] = (int32_t)(s->srec.size() / 2);
            s->sdesc[(size_t)2 * g + 1] = (int32_t)sload[(size_t)g];
            s->max_stream_steps = std::max<int32_t>(s->max_stream_steps, (int32_t)sload[(size_t)g]);
            // Order inside a stream: by row; in the update-beside-the-walk schedules the LONGEST ROW LAST -- every stream ends with a
            // flush, and the fewer other flushes fall into the walk's last few thousand cycles, the shorter the queue the update
            // waves are left with when the walkers are gone (tools/step_probe.py, one batch at a time / two in flight: by row
            // 0.6446 / 0.5727 ms, longest last 0.6415 / 0.5664, shortest last 0.6436 / 0.5754).  ULTRA_STREAM_ROW_ORDER=0 / 1 / 2
            // (by row / shortest last / longest last) overrides for measurements.  The sums do not depend on it.
            static const int row_order_env = std::getenv("ULTRA_STREAM_ROW_ORDER") ? std::atoi(std::getenv("ULTRA_STREAM_ROW_ORDER")) : -1;
            const int row_order = row_order_env >= 0 ? row_order_env : (walkers == 12 ? 2 : 0);
            std::sort(srows[(size_t)g].begin(), srows[(size_t)g].end(), [&](int32_t a, int32_t b) {
                const Item &ia = p->items[(size_t)a], &ib = p->items[(size_t)b];
                if (row_order == 1 && ia.len != ib.len) return ia.len > ib.len;
                if (row_order == 2 && ia.len != ib.len) return ia.len < ib.len;
                return ia.row < ib.row;
            });
            // (twelve-walker schedules: records pre-multiplied by the 256-byte row pitch, plan.hpp ULTRA_STREAM_PRESHIFT; the
            // products fit 32 bits: the caller keeps node and relation counts below 2^24)
            const int shift = (walkers == 12 && ULTRA_STREAM_PRESHIFT) ? 8 : 0;
            s->rec_shift = shift;
            for (int32_t it : srows[(size_t)g]) {
                const Item &row = p->items[(size_t)it];
                for (int32_t e = row.begin; e < row.begin + row.len; ++e) {
                    s->srec.push_back((int32_t)((uint32_t)p->col[(size_t)e] << shift));
                    s->srec.push_back((int32_t)((uint32_t)p->type[(size_t)e] << shift));
                }
                s->srec.push_back((int32_t)((uint32_t)row.row << shift));     // marker: flush the accumulator to this row
                s->srec.push_back((int32_t)((uint32_t)p->num_rel << shift));
            }
        }
        // (records are requested two rounds ahead without a bounds test, and the four streams of a wave are walked for as many
        // rounds as the longest of them: the last wave's shorter streams request up to that many records past their own end)
        s->srec_pad = 2 * ((int64_t)ORDER_PAD + (int64_t)s->max_stream_steps);
        s->srec.resize(s->srec.size() + (size_t)s->srec_pad, 0);
        // rows per workgroup: chain rows + stream rows, ascending, whole 32-row tiles
        s->prow_ptr.assign((size_t)nparts + 1, 0);
        for (int32_t q = 0; q < nparts; ++q) {
            std::vector<int32_t> mine;
            for (int32_t c : part_chain[(size_t)q]) mine.push_back(p->items[(size_t)c].row);
            for (int g = 0; g < ORDER_GROUPS; ++g)
                for (int32_t it : srows[(size_t)q * ORDER_GROUPS + g]) mine.push_back(p->items[(size_t)it].row);
            std::sort(mine.begin(), mine.end());
            s->prow.insert(s->prow.end(), mine.begin(), mine.end());
            while (s->prow.size() % 32) s->prow.push_back(-1);
            s->prow_ptr[(size_t)q + 1] = (int32_t)s->prow.size();
            s->max_rows = std::max<int32_t>(s->max_rows, (int32_t)mine.size());
            s->max_chain_rows = std::max<int32_t>(s->max_chain_rows, (int32_t)part_chain[(size_t)q].size());
        }
    }
    // the chain producers request descriptors a fixed number of chunks ahead without a bounds test: readable, harmless
    // entries (edge 0) behind the last chunk
    for (int k = 0; k < CHUNK_PAD; ++k) s->chunks.push_back(Chunk{0, 0, 0, 0});
    return s;
}

}  // namespace ultra

using ultra::set_error;

extern "C" {

int32_t ultra_abi_version(void) { return ULTRA_ABI_VERSION; }

const char *ultra_last_error(void) { return ultra::g_last_error.c_str(); }

int32_t ultra_plan_create(ultra_plan **plan, const int64_t *edge_index, const int64_t *edge_type, int64_t E,
                          int64_t num_out, int64_t num_in, int64_t num_rel, const ultra_plan_opts *opts) {
    if (!plan) {
        set_error("ultra_plan_create: plan is NULL");
        return ULTRA_ERR_INVALID;
    }
    *plan = nullptr;
    if (E < 0 || num_out < 0 || num_in < 0 || num_rel < 0 || (E > 0 && (!edge_index || !edge_type))) {
        set_error("ultra_plan_create: negative size or NULL edge arrays");
        return ULTRA_ERR_INVALID;
    }
    if (E >= (int64_t(1) << 31) || num_out >= (int64_t(1) << 31) || num_in >= (int64_t(1) << 31) ||
        num_rel >= (int64_t(1) << 31)) {
        set_error("ultra_plan_create: sizes beyond int32 are not supported");
        return ULTRA_ERR_UNSUPPORTED;
    }
    std::vector<int32_t> row((size_t)E), col((size_t)E), type((size_t)E);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t r = edge_index[e], c = edge_index[E + e], t = edge_type[e];
        if (r < 0 || r >= num_out || c < 0 || c >= num_in || t < 0 || t >= num_rel) {
            set_error("ultra_plan_create: edge " + std::to_string(e) + " out of range (row " + std::to_string(r) +
                      ", col " + std::to_string(c) + ", type " + std::to_string(t) + ")");
            return ULTRA_ERR_INVALID;
        }
        row[(size_t)e] = (int32_t)r;
        col[(size_t)e] = (int32_t)c;
        type[(size_t)e] = (int32_t)t;
    }
    if (opts && (opts->flags & ULTRA_PLAN_DENSE)) {
        const int64_t cells = ((num_out + 31) / 32) * 32 * ((num_in + 7) / 8 + ULTRA_DENSE_KG_ALIGN) * 8 * (num_rel + 3);
        if (num_in > ULTRA_DENSE_MAX_IN_ROW || cells > (int64_t(1) << 28)) {
            set_error("ultra_plan_create: ULTRA_PLAN_DENSE needs num_in_row <= " + std::to_string(ULTRA_DENSE_MAX_IN_ROW) +
                      " and at most 2^28 adjacency cells");
            return ULTRA_ERR_UNSUPPORTED;
        }
    }
    *plan = ultra::build_plan(row.data(), col.data(), type.data(), E, num_out, num_in, num_rel, opts, true);
    if ((*plan)->dense_overflow) {
        delete *plan;
        *plan = nullptr;
        set_error("ultra_plan_create: ULTRA_PLAN_DENSE stores multiplicities as bytes; an edge is repeated more than 255 times");
        return ULTRA_ERR_UNSUPPORTED;
    }
    return ULTRA_OK;
}

int32_t ultra_plan_get_info(const ultra_plan *p, ultra_plan_info *info) {
    if (!p || !info) {
        set_error("ultra_plan_get_info: NULL argument");
        return ULTRA_ERR_INVALID;
    }
    info->num_edge = p->num_edge;
    info->num_node = p->num_out;
    info->num_relation = p->num_rel;
    info->n_item = (int64_t)p->items.size();
    info->n_wave_item = p->n_w;
    info->n_group_item = p->n_g;
    info->n_unit = p->n_unit;
    info->n_split_row = (int64_t)p->split_row.size();
    info->n_partial_slot = p->n_slot;
    info->seg_len = p->seg_len;
    info->g_max = p->g_max;
    info->flags = p->flags;
    info->packed = p->packed_ok ? 1 : 0;
    info->on_device = p->on_device ? 1 : 0;
    info->has_transpose = (p->tplan && p->rplan) ? 1 : 0;
    info->n_type_run = p->n_type_run;
    info->dense_bytes = (int64_t)p->a_frag.size();
    info->n_chain_row = p->n_chain;
    info->dense_order_bytes = (int64_t)p->a_ex.size();
    return ULTRA_OK;
}

int32_t ultra_plan_export(const ultra_plan *p, int32_t which, void *dst, int64_t capacity, int64_t *count) {
    if (!p || !count) {
        set_error("ultra_plan_export: NULL argument");
        return ULTRA_ERR_INVALID;
    }
    const void *src = nullptr;
    int64_t n = 0;
    switch (which) {
        case ULTRA_ARR_ROW_PTR: src = p->row_ptr.data(); n = (int64_t)p->row_ptr.size(); break;
        case ULTRA_ARR_COL: src = p->col.data(); n = (int64_t)p->col.size(); break;
        case ULTRA_ARR_TYPE: src = p->type.data(); n = (int64_t)p->type.size(); break;
        case ULTRA_ARR_PERM: src = p->perm.data(); n = (int64_t)p->perm.size(); break;
        case ULTRA_ARR_ITEM: src = p->items.data(); n = (int64_t)p->items.size() * 4; break;
        case ULTRA_ARR_SPLIT_ROW: src = p->split_row.data(); n = (int64_t)p->split_row.size(); break;
        case ULTRA_ARR_SPLIT_PTR: src = p->split_ptr.data(); n = (int64_t)p->split_ptr.size(); break;
        case ULTRA_ARR_DENSE: src = p->a_frag.data(); n = (int64_t)p->a_frag.size() / 4; break;
        case ULTRA_ARR_DENSE_ORDER: src = p->a_ex.data(); n = (int64_t)p->a_ex.size() / 4; break;
        default: set_error("ultra_plan_export: unknown array id"); return ULTRA_ERR_INVALID;
    }
    *count = n;
    if (dst) {
        if (capacity < n) {
            set_error("ultra_plan_export: destination too small");
            return ULTRA_ERR_INVALID;
        }
        if (n) std::memcpy(dst, src, (size_t)n * sizeof(int32_t));
    }
    return ULTRA_OK;
}

}  // extern "C"
