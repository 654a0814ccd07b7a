"""Generator of ultra_amd/csrc/rspmm_order_asm.hpp: the two software-pipelined inner loops of the reference-order
rspmm kernel (fp32, unit weights, relation slice in LDS) as gfx950 assembly.

    python tools/gen_order_asm.py            # rewrites ultra_amd/csrc/rspmm_order_asm.hpp

Why assembly.  Both loops keep vector-memory loads in flight ACROSS loop iterations (source rows requested one or two
chunks before they are summed).  hipcc handles that badly in two ways (ROCm 7.2, read off the ISA): its vmcnt
bookkeeping gives up at the back-edge (`s_waitcnt vmcnt(0)` at the top of every round: the queue is drained before the
next requests go out) and it rotates loop-carried load destinations through register copies, each of which waits for
the load it copies.  Issuing the loads as `asm volatile` from C++ does not help: the compiler then copies (or spills)
registers whose load has not landed.  Inside one asm statement the registers, the issue order and every wait count are
fixed by this file:

  * loads of one wave return in issue order, so `s_waitcnt vmcnt(N)` means "everything but the N youngest requests has
    landed"; every count below is derived from the issue sequence noted beside it;
  * all in-flight destinations are clobber registers of the statement (v64..v127): the compiler keeps nothing there
    across it, and every path out of the statement ends in vmcnt(0);
  * the arithmetic is the C++ path's: v_pk_mul_f32 / v_pk_add_f32 (or v_min / v_max) per message in sorted edge
    order, one accumulator per output element -- bit-identical to rspmm.cpp:61-72.

The generated header is committed; this script is its source.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("ULTRA_GEN_OUT") or os.path.join(os.path.dirname(HERE), "ultra_amd", "csrc", "rspmm_order_asm.hpp")

BINOPS = {0: "v_pk_mul_f32", 1: "v_pk_add_f32"}            # BIN_MUL, BIN_ADD (operator.cuh:15,29)
IDENT = {0: "0", 1: "0x7f7fffff", 2: "0xff7fffff"}
# cache policy of the record streams: default.  " nt" (streaming) was measured slower: loads return in order, so the
# longer latency of an nt request stands in front of the gathers queued behind it.
REC_POLICY = os.environ.get("ULTRA_GEN_REC_POLICY", "")
# cache policy of the flush stores of the stream walk: nt (written once, read by the NEXT kernel).  Measured (r3, rocprofv3
# FETCH_SIZE + WRITE_SIZE): FB15k237 bs 8 180 -> 155 MB and 78.3 -> 77.5 us, CoDEx-L bs 8 1.91 -> 1.88 GB and 266 -> 257 us;
# " sc1" the same within noise; "" = default policy.
OUT_POLICY = os.environ.get("ULTRA_GEN_OUT_POLICY", " nt")


def vr(lo, n=1):
    return "v%d" % lo if n == 1 else "v[%d:%d]" % (lo, lo + n - 1)


class Asm:
    def __init__(self):
        self.lines = []

    def __call__(self, text, comment=None):
        self.lines.append((text, comment))

    def label(self, name):
        self.lines.append((name + ":", None))

    def render(self, indent="        "):
        out = []
        for text, comment in self.lines:
            c = ("   // " + comment) if comment else ""
            out.append('%s"%s\\n"%s' % (indent, text, c))
        return "\n".join(out)


def nary(a, sum_code, acc, x):
    """acc[0..3] (+)= x[0..3] under the current exec mask"""
    if sum_code == 0:
        a("v_pk_add_f32 %s, %s, %s" % (vr(acc, 2), vr(acc, 2), vr(x, 2)))
        a("v_pk_add_f32 %s, %s, %s" % (vr(acc + 2, 2), vr(acc + 2, 2), vr(x + 2, 2)))
    else:
        op = "v_min_f32_e32" if sum_code == 1 else "v_max_f32_e32"
        for e in range(4):
            a("%s %s, %s, %s" % (op, vr(acc + e), vr(acc + e), vr(x + e)))


# ---------------------------------------------------------------------------------------------------------------
# group streams: every 16-lane group of a workgroup walks ONE stream -- the rows the schedule gave it, laid out back
# to back in the stream's own record array: a row's edges in sorted order as (col, type) records, closed by a marker
# record (row, num_rel).  A marker step flushes: (+ boundary), store, accumulator back to the identity.  So the walk is
# one continuous software pipeline over the whole stream -- no per-row restart (two dependent memory round trips per
# four rows in the unit walk), no steps masked because the four rows of a unit differ in length, no per-unit item
# loads.  Two 4-step chunks of source rows + one round (8 steps) of records are in flight per lane.
# Register map (clobbers):
#   v64..v79   chunk A source rows (4 x 16 B)    v96..v99    LDS addresses of chunk A's relation rows (marker: row num_rel)
#   v80..v95   chunk B source rows               v100..v103  ... of chunk B's
#   v104..v107 / v108..v111  byte offsets node * row_bytes + lane_bytes of chunk A / B: the gather offset of an edge, the
#                            store offset of a marker (source and output matrix have the same row stride)
#   v112:113   records of the round after the current one (in flight)      v120:121  round 0's records (prologue)
#   v114/v115  current round's col / type (lane l holds step l % 8)
#   v56..v63, v48..v55  the chunk's four relation rows          v116..v119 the accumulator       v122 scratch
#   POST == 2 (the walk parks its finished rows in LDS for the workgroup's update waves, see stream_park):
#   v123 ring slot     v124 LDS address of the control block     v125 the constant 1     v126 scratch
STREAM_CLOBBER_LO, STREAM_CLOBBER_HI = 48, 122
RV = (56, 60, 48, 52)
ACC = 116


# ---- the hand-off (POST == 2): the flushed row itself goes to the update waves through LDS ----
# (POST == 1 -- rows stored to memory and their offsets queued in LDS, round 4's by-reference form -- was removed in round 5)
# Tiles of 16 rows x (256 + 16 pad) bytes, HANDOFF2_NT of them, reused round-robin by "generations" of 16 rows: slot g (from
# the atomic tail) belongs to generation g / 16, tile buffer (g / 16) % NT, row g % 16.  Control block (byte offsets behind
# qctl; rspmm_order_kernels.hpp has the same numbers): 0 tail, 4 walkers done, 8 update-wave barrier, 12 chain done,
# 16 update waves done with a block (two generations; four per block), 32 + 4 buf: rows posted into buffer buf (monotonic: generation G is complete at 16 (G / NT + 1)),
# 64 + 4 i: byte offset of the node row parked in tile row i (i < 16 NT).
HANDOFF2_NT = 4
HANDOFF2_ROW_BYTES = 272
HANDOFF2_CONSUMED_OFF, HANDOFF2_POSTED_OFF, HANDOFF2_ROWID_OFF = 16, 32, 64
# a park wait that gave up leaves a non-zero word here (UPD2_CTL_ERR in update_tile.hpp; the kernel's C++ forwards it to the host)
HANDOFF2_ERR_OFF = 28
HANDOFF2_RETRY_OFF = 24     # (UPD2_CTL_RETRIES: how often a park wait found the ring full and went to sleep)
PARK_DIAG = os.environ.get("ULTRA_GEN_PARK_DIAG", "0") == "1"     # measurement builds: count those retries (trace hook)
PARK_SPIN_CAP = 1 << 20
HANDOFF2_X_OFF = HANDOFF2_NT * 16 * HANDOFF2_ROW_BYTES     # the x rows' ring sits behind the aggregates'


def stream_park(a, ob_q, x_q, tag):
    """Flush of the hand-off (POST == 2), for the lanes in %[mk] (the 16-lane groups at a marker): accumulator (boundary already applied) ->
    a row of the current tile, and beside it the row's x (registers x_q..: a marker step gathers at its own row's offset).  One lane per group takes the slot and later writes the row's id and bumps the buffer's
    count; all 16 write their 16 bytes.  A slot of generation G may be written once generation G - NT has been consumed
    (the update waves never wait for a walker that waits for them: the slots of the NT generations they may be working on
    are all writable).  LDS operations of a wave execute in order: who sees the count sees the row."""
    a("v_cmp_eq_u32_e32 vcc, 0, %[lb]", "lane 0 of each group (whole-span rows: lb = 16 (lane % 16))")
    a("s_and_b64 exec, %[mk], vcc")
    a("ds_add_rtn_u32 v123, v124, v125", "slot = tail++")
    a("s_mov_b64 exec, %[mk]")
    a("ds_read_b32 v126, v124 offset:%d" % HANDOFF2_CONSUMED_OFF, "(with the slot: ONE LDS round trip per flush -- they cost hundreds of cycles here)")
    a("s_waitcnt lgkmcnt(0)")
    a("s_nop 4")
    a("v_mov_b32_dpp v123, v123 row_newbcast:0 row_mask:0xf bank_mask:0xf", "the group's slot to its 16 lanes, in registers")
    a("v_lshrrev_b32_e32 v122, 4, v123", "generation")
    a("s_mov_b32 %[t1], 0", "polls of this wait (bounded: OrderParams::err)")
    a.label(".Lpark_wait_%s_%%=" % tag)
    a("v_lshrrev_b32_e32 v126, 2, v126", "(the word counts the update waves that are done with a block of two generations: four per block)")
    a("v_lshl_add_u32 v126, v126, 1, %d" % HANDOFF2_NT)
    a("v_cmp_gt_u32_e32 vcc, v126, v122", "consumed + NT > generation: the buffer is free")
    a("s_andn2_b64 vcc, exec, vcc")
    a("s_cbranch_vccz .Lpark_go_%s_%%=" % tag)
    a("s_sleep 2")
    if PARK_DIAG:      # (measurement builds: one count per retry and wave -- an LDS atomic by every parking lane costs the walk 50 %)
        a("s_mov_b64 %[mk], exec")
        a("s_mov_b64 exec, 1")
        a("ds_add_u32 v124, v125 offset:%d" % HANDOFF2_RETRY_OFF)
        a("s_mov_b64 exec, %[mk]")
    a("s_add_u32 %[t1], %[t1], 1")
    a("s_cmp_lt_u32 %%[t1], 0x%x" % PARK_SPIN_CAP)
    a("s_cbranch_scc0 .Lpark_giveup_%s_%%=" % tag)
    a("ds_read_b32 v126, v124 offset:%d" % HANDOFF2_CONSUMED_OFF)
    a("s_waitcnt lgkmcnt(0)")
    a("s_branch .Lpark_wait_%s_%%=" % tag)
    a.label(".Lpark_giveup_%s_%%=" % tag)
    # the update waves have not freed a ring row in 2^20 polls (>= 0.1 s; a block takes them microseconds): leave the code for the
    # kernel's C++ to report and park anyway -- the launch ends with an error word instead of hanging the GPU
    a("ds_write_b32 v124, v125 offset:%d" % HANDOFF2_ERR_OFF)
    a.label(".Lpark_go_%s_%%=" % tag)
    a("v_and_b32_e32 v122, %d, v123" % (16 * HANDOFF2_NT - 1), "tile row among the NT buffers")
    a("v_mad_u32_u24 v126, v122, %[rowpitch], %[lb]")
    a("v_add_u32_e32 v126, %[qtile], v126")
    a("ds_write_b128 v126, %s" % vr(ACC, 4))
    a("ds_write_b128 v126, %s offset:%d" % (vr(x_q, 4), HANDOFF2_X_OFF), "... and x[row]: what the marker's gather brought")
    a("v_cmp_eq_u32_e32 vcc, 0, %[lb]")
    a("s_and_b64 exec, %[mk], vcc")
    a("v_lshl_add_u32 v126, v122, 2, v124")
    a("ds_write_b32 v126, v%d offset:%d" % (ob_q, HANDOFF2_ROWID_OFF), "which node row it is (byte offset)")
    a("v_lshrrev_b32_e32 v122, 4, v122")
    a("v_lshl_add_u32 v126, v122, 2, v124")
    a("ds_add_u32 v126, v125 offset:%d" % HANDOFF2_POSTED_OFF)
    a("s_mov_b64 exec, %[mk]")
    # (no wait: LDS instructions take their operands at issue, and they complete in order -- the counted waits of the steps
    # below only ever wait longer for it)


# How a step's (col, type) gets from the lane that holds the record to the 16 lanes of its group: DPP row broadcast (a VALU move:
# "row_newbcast" is the assembler's name of row_share on gfx90a+) instead of ds_swizzle -- the swizzles were 8 of the 12 LDS
# instructions of a chunk, and the LDS pipe of the CU, shared by all sixteen waves, is what the walk keeps busiest.
DPP_BCAST = os.environ.get("ULTRA_GEN_DPP_BCAST", "1") == "1"
# The walk's instruction diet (round 5; ULTRA_GEN_DIET=0 generates round 4's loops for A/B builds).  The stream phase shares its
# SIMDs with the update waves' matrix chains, and what it loses to them is VALU issue slots: 9.4 VALU instructions per step
# before, 6.5 now.
#   * marker test once per ROUND of records instead of once per chunk of gathers: at promote time lane l holds the type of step
#     l % 8, so ONE v_cmp against the marker's type gives the 64-lane mask of marker steps; folded to 32 bits (vcc_lo | vcc_hi) its
#     nibbles say "chunk A (steps 0..3) / chunk B (steps 4..7) of some group holds a marker" -- two SALU tests where v_max3 +
#     v_max + v_cmp ran per chunk;
#   * live test against the step counter in an SGPR (lim = len - lane % 8 and len never change) instead of a per-lane remaining
#     count that a v_add moved every round;
#   * PRESHIFT (the twelve-walker schedules' records, POST variants): records hold col * 256 and type * 256 (whole-span rows of
#     256 bytes: the only geometry the hand-off forms serve), so a step's gather offset and relation-row address are ONE VOP2 add
#     each with the DPP row broadcast ON ITS SOURCE -- v_add_u32_dpp dst, record, lane_base row_newbcast:k -- where a DPP move
#     plus a v_mad_u32_u24 / v_lshl_add_u32 (VOP3: no DPP on gfx950) ran before: 8 VALU instead of 16 per chunk.
DIET = os.environ.get("ULTRA_GEN_DIET", "1") == "1"


def stream_fetch(a, xb, tb, ob, J, preshift=False):
    if preshift:
        a("s_nop 1", "(a VALU write of v114 / v115 needs two wait states before a DPP read)")
        for q in range(4):
            a("v_add_u32_dpp v%d, v114, %%[lb] row_newbcast:%d row_mask:0xf bank_mask:0xf" % (ob + q, J + q),
              "col * 256 of step %d, broadcast inside the group, + the lane's 16 bytes" % (J + q))
        for q in range(4):
            a("v_add_u32_dpp v%d, v115, %%[lds] row_newbcast:%d row_mask:0xf bank_mask:0xf" % (tb + q, J + q))
        for q in range(4):
            a("global_load_dwordx4 %s, v%d, %%[xb]" % (vr(xb + 4 * q, 4), ob + q))
        return
    if DPP_BCAST:
        a("s_nop 1", "(a VALU write of v114 / v115 needs two wait states before a DPP read)")
        for q in range(4):
            a("v_mov_b32_dpp v%d, v114 row_newbcast:%d row_mask:0xf bank_mask:0xf" % (ob + q, J + q))
        for q in range(4):
            a("v_mov_b32_dpp v%d, v115 row_newbcast:%d row_mask:0xf bank_mask:0xf" % (tb + q, J + q))
        for q in range(4):
            a("v_mad_u32_u24 v%d, v%d, %%[xrb], %%[lb]" % (ob + q, ob + q))
            a("global_load_dwordx4 %s, v%d, %%[xb]" % (vr(xb + 4 * q, 4), ob + q))
        for q in range(4):
            a("v_lshl_add_u32 v%d, v%d, 8, %%[lds]" % (tb + q, tb + q))
        return
    for q in range(4):
        a("ds_swizzle_b32 v%d, v114 offset:swizzle(BROADCAST,16,%d)" % (ob + q, J + q))
    for q in range(4):
        a("ds_swizzle_b32 v%d, v115 offset:swizzle(BROADCAST,16,%d)" % (tb + q, J + q))
    for q in range(4):
        a("s_waitcnt lgkmcnt(%d)" % (7 - q))
        a("v_mad_u32_u24 v%d, v%d, %%[xrb], %%[lb]" % (ob + q, ob + q))
        a("global_load_dwordx4 %s, v%d, %%[xb]" % (vr(xb + 4 * q, 4), ob + q))
    a("s_waitcnt lgkmcnt(0)")
    for q in range(4):
        a("v_lshl_add_u32 v%d, v%d, 8, %%[lds]" % (tb + q, tb + q))


def stream_rel_reads(a, tb):
    """the chunk's four relation rows: requested BEFORE the wait for its source rows, so that they land under it"""
    for q in range(4):
        a("ds_read_b128 %s, v%d" % (vr(RV[q], 4), tb + q))


def stream_compute(a, xb, tb, ob, consts, binop, sum_code, first_step, tag, post=0):
    """The chunk's four steps in order.  Fast form: every stream of the wave still has the whole chunk (uniform test
    against nf) and none of the 16 (group, step) slots is a marker (the marker's LDS address is the largest there is).
    General form, per step: live = consts[q] < rem; marker = live and relation address == mark; edges accumulate in the
    lanes live & ~marker, markers flush.
    DIET: t0 holds kb + first_step + 4 wherever a chunk is computed; the marker test is the round's mask (m32: chunk A's nibbles,
    mb: chunk B's, masked out before the next round's promote); live = kb + first_step + q < len with the step in an SGPR."""
    if DIET:
        a("s_cmp_le_i32 %[t0], %[nf]")
        a("s_cbranch_scc0 .Lstream_general_%s_%%=" % tag)
        if first_step == 0:
            a("s_and_b32 %[t1], %[m32], 0x0f0f0f0f", "a marker among steps 0..3 of some group?")
        else:
            a("s_cmp_lg_u32 %[mb], 0", "... among steps 4..7?")
        a("s_cbranch_scc1 .Lstream_general_%s_%%=" % tag)
    else:
        a("s_add_i32 %%[t1], %%[kb], %d" % (first_step + 4))
        a("s_cmp_le_i32 %[t1], %[nf]")
        a("s_cbranch_scc0 .Lstream_general_%s_%%=" % tag)
        a("v_max3_u32 v122, v%d, v%d, v%d" % (tb, tb + 1, tb + 2))
        a("v_max_u32_e32 v122, v122, v%d" % (tb + 3))
        a("v_cmp_eq_u32_e32 vcc, v122, %[mark]")
        a("s_cbranch_vccnz .Lstream_general_%s_%%=" % tag)
    for q in range(4):
        a("s_waitcnt lgkmcnt(%d)" % (3 - q))
        x = xb + 4 * q
        a("%s %s, %s, %s" % (binop, vr(x, 2), vr(RV[q], 2), vr(x, 2)))
        a("%s %s, %s, %s" % (binop, vr(x + 2, 2), vr(RV[q] + 2, 2), vr(x + 2, 2)))
        nary(a, sum_code, ACC, x)
    a("s_branch .Lstream_summed_%s_%%=" % tag)
    a.label(".Lstream_general_%s_%%=" % tag)
    for q in range(4):
        a("s_waitcnt lgkmcnt(%d)" % (3 - q))
        x = xb + 4 * q
        if post != 2:
            a("%s %s, %s, %s" % (binop, vr(x, 2), vr(RV[q], 2), vr(x, 2)))
            a("%s %s, %s, %s" % (binop, vr(x + 2, 2), vr(RV[q] + 2, 2), vr(x + 2, 2)))
        if DIET:
            a("s_add_i32 %%[t1], %%[kb], %d" % (first_step + q))
            a("v_cmp_lt_i32_e32 vcc, %[t1], %[rem]", "live: this step is inside the group's stream")
        else:
            a("v_cmp_lt_i32_e32 vcc, %d, %%[rem]" % consts[q], "live")
        a("v_cmp_eq_u32_e64 %%[mk], v%d, %%[mark]" % (tb + q))
        a("s_and_b64 %[mk], %[mk], vcc", "marker")
        a("s_andn2_b64 exec, vcc, %[mk]", "edges")
        if post == 2:
            # (a marker's gather offset is its own row's: its lanes hold x[row] -- left as it is, parked with the aggregate)
            a("%s %s, %s, %s" % (binop, vr(x, 2), vr(RV[q], 2), vr(x, 2)))
            a("%s %s, %s, %s" % (binop, vr(x + 2, 2), vr(RV[q] + 2, 2), vr(x + 2, 2)))
        nary(a, sum_code, ACC, x)
        a("s_mov_b64 exec, %[mk]")
        a("s_cbranch_execz .Lstream_noflush_%s%d_%%=" % (tag, q))
        # flush: out[row] = acc (+) boundary[row]   (rspmm.cpp:70-72), then a fresh accumulator
        a("v_cmp_eq_u32_e32 vcc, v%d, %%[bndoff]" % (ob + q))
        if sum_code == 0:
            a("s_and_b64 exec, exec, vcc", "lanes of the boundary row")
            for e in range(4):
                a("v_add_f32_e32 v%d, v%d, %%[b%d]" % (ACC + e, ACC + e, e))
            a("s_mov_b64 exec, %[mk]")
        else:
            # min / max: the boundary row meets its value, every other row the fill (layers.py:206-207: the boundary
            # tensor is zero off the query rows -> fill 0; a call without that tensor passes -+inf, which changes nothing)
            op = {1: "v_min_f32_e32", 2: "v_max_f32_e32"}[sum_code]
            for e in range(4):
                a("v_cndmask_b32_e32 v122, %%[bz], %%[b%d], vcc" % e)
                a("%s v%d, v%d, v122" % (op, ACC + e, ACC + e))
        if post == 2:
            stream_park(a, ob + q, xb + 4 * q, "%s%d" % (tag, q))
        else:
            a("global_store_dwordx4 v%d, %s, %%[ob]%s" % (ob + q, vr(ACC, 4), OUT_POLICY))
        a("s_nop 2", "gfx940+: a VALU write of the data registers of a > 8-byte store needs 2 wait states behind the store "
                     "(with one, v_mov v116 reached the last lanes' data first)")
        for e in range(4):
            a("v_mov_b32_e32 v%d, %s" % (ACC + e, IDENT[sum_code]))
        a.label(".Lstream_noflush_%s%d_%%=" % (tag, q))
        a("s_mov_b64 exec, %[ex]")
    a.label(".Lstream_summed_%s_%%=" % tag)


def gen_stream(sum_code, mul_code, rec_policy, post=0):
    a = Asm()
    binop = BINOPS[mul_code]
    A, B, TA, TB, OA, OB = 64, 80, 96, 100, 104, 108
    preshift = DIET and post != 0

    def promote(rec, step_reg=None):
        if DIET:
            # (%[l8] carries lim = len - lane % 8 in these variants: step kr + lane % 8 is inside the stream iff kr < lim)
            if step_reg is None:
                a("v_cmp_lt_i32_e32 vcc, 0, %[l8]")
            else:
                a("v_cmp_lt_i32_e32 vcc, %%[%s], %%[l8]" % step_reg)
        else:
            a("v_cmp_lt_i32_e32 vcc, %[l8], %[rem]")
        a("v_cndmask_b32_e32 v114, 0, v%d, vcc" % rec, "steps past the stream's end gather node 0 ...")
        a("v_cndmask_b32_e32 v115, 0, v%d, vcc" % (rec + 1), "... multiply by relation 0, are no marker, and are masked out of the sum")
        if DIET:
            a("v_cmp_eq_u32_e32 vcc, %[rmk], v115", "marker steps of this round: lane l = step l % 8 of its group")
            a("s_or_b32 %[m32], vcc_lo, vcc_hi", "(VALU -> SGPR -> SALU: interlocked)")

    def compute(xb, tb, ob, consts, first_step, tag):
        stream_compute(a, xb, tb, ob, consts, binop, sum_code, first_step, tag, post)

    a("s_mov_b64 %[ex], exec")
    a("s_mov_b32 %[kb], 0")
    for e in range(4):
        a("v_mov_b32_e32 v%d, %s" % (ACC + e, IDENT[sum_code]))
    if post:
        a("v_mov_b32_e32 v124, %[qctl]")
        a("v_mov_b32_e32 v125, 1")
    a("global_load_dwordx2 v[120:121], %%[roff], %%[rb]%s" % rec_policy, "records of round 0")
    a("global_load_dwordx2 v[112:113], %%[roff], %%[rb] offset:64%s" % rec_policy, "records of round 1")
    a("v_add_u32_e32 %[roff], 0x80, %[roff]")
    a("s_waitcnt vmcnt(1)", "in flight: [r0, r1] -> r0")
    promote(120)
    stream_fetch(a, A, TA, OA, 0, preshift)                        # in flight: [r1, A x 4]
    a.label(".Lstream_loop_%=")
    a("s_add_i32 %[t0], %[kb], 4")
    a("s_cmp_lt_i32 %[t0], %[ns]")
    a("s_cbranch_scc0 .Lstream_last_a_%=")
    stream_fetch(a, B, TB, OB, 4, preshift)                        # [r', A x 4, B x 4]
    stream_rel_reads(a, TA)
    a("s_waitcnt vmcnt(4)", "[r', A x 4, B x 4] (+ flush stores, which only make the count stricter) -> r', A; the previous "
                            "chunk's flush stores are older than B x 4: complete")
    compute(A, TA, OA, (0, 1, 2, 3), 0, "a")
    if DIET:
        a("s_and_b32 %[mb], %[m32], 0xf0f0f0f0", "this round's markers among steps 4..7 (the next promote overwrites m32)")
    a("s_add_i32 %[t0], %[kb], 8")
    a("s_cmp_lt_i32 %[t0], %[ns]")
    a("s_cbranch_scc0 .Lstream_last_b_%=")
    if not DIET:
        a("v_add_u32_e32 %[rem], -8, %[rem]", "rem = len - (kb + 8)")
    promote(112, "t0")
    a("global_load_dwordx2 v[112:113], %%[roff], %%[rb]%s" % rec_policy, "records of round kb / 8 + 2: a whole round before their first use")
    a("v_add_u32_e32 %[roff], 64, %[roff]")
    stream_fetch(a, A, TA, OA, 0, preshift)                        # [B x 4, r'', A x 4]
    stream_rel_reads(a, TB)
    a("s_waitcnt vmcnt(5)", "[B x 4, r'', A x 4] -> B; chunk A's flush stores are older than r'', A x 4: complete")
    compute(B, TB, OB, (-4, -3, -2, -1), 4, "b")   # (rem already moved on by 8)
    a("s_mov_b32 %[kb], %[t0]")
    a("s_branch .Lstream_loop_%=")
    a.label(".Lstream_last_a_%=")
    stream_rel_reads(a, TA)
    a("s_waitcnt vmcnt(0)")
    compute(A, TA, OA, (0, 1, 2, 3), 0, "la")
    a("s_branch .Lstream_done_%=")
    a.label(".Lstream_last_b_%=")
    stream_rel_reads(a, TB)
    a("s_waitcnt vmcnt(0)")
    compute(B, TB, OB, (4, 5, 6, 7), 4, "lb")
    a.label(".Lstream_done_%=")
    a("s_waitcnt vmcnt(0)", "flush stores")
    return a


# ---------------------------------------------------------------------------------------------------------------
# chain producers: 16-lane group `slot` of waves 1..15 computes message `slot` of every 60-edge chunk of the workgroup's
# chunk list and parks it (transposed, see CHAIN_QUADS in rspmm_order_kernels.hpp) in the ring half of the chunk.
# Per lane and chunk k three requests go through the vector-memory queue, D chunks apart: B_k (the chunk's first edge,
# from its descriptor), R_k (this slot's record) and G_k (the record's source row).  Register map (clobbers):
#   v64..v95   xq[j]  source row of the chunk in stage j        v96..v111  rq[j]  record (col, type) of the chunk D further on
#   v112..v119 tq[j]  LDS address of the relation row of xq[j]  v56..v63   bq[j]  first edge of the chunk 2 D further on
#   v120..v123, v52..v55, v48..v51, v44..v47  relation row -> message -> transposed message, one set per chunk, rotating:
#                          while message i is parked, message i + 1 is multiplied and transposed and the relation row
#                          of chunk i + 2 is on its way
#   v124, v125 gather / record offsets    v126 ring address of this chunk    v127 descriptor offset of the last B request
# Descriptors are read with vector loads on purpose: a scalar load shares lgkmcnt with the LDS traffic and returns out of
# order, so every LDS wait behind it would have to be lgkmcnt(0) -- one scalar-cache round trip per chunk on the
# critical path (measured: 800 cycles per chunk with per-chunk s_loads).  The descriptor list is readable CHUNK_PAD
# (>= 3 D) entries past its end (plan.cpp), so no index is clamped; a prefetch past the end requests edge 0 or a
# neighbouring workgroup's chunk -- valid addresses whose data nobody consumes.
PROD_D = 8
PROD_CLOBBER_LO, PROD_CLOBBER_HI = 44, 127
# (ULTRA_GEN_WAVES: waves per workgroup of the build the header is generated for -- plan.hpp ULTRA_ORDER_WAVES)
GEN_WAVES = int(os.environ.get("ULTRA_GEN_WAVES", "16"))
RING_HALF_BYTES = (GEN_WAVES - 1) * 64 * 16


def gen_producer(mul_code, rec_policy):
    a = Asm()
    D = PROD_D
    binop = BINOPS[mul_code]
    SETS = (120, 52, 48, 44)   # four 4-register sets: relation row -> message -> transposed message, rotating per chunk

    def xq(j):
        return 64 + 4 * (j % D)

    def rq(j):
        return 96 + 2 * (j % D)

    def tq(j):
        return 112 + (j % D)

    def bq(j):
        return 56 + (j % D)

    def begin_request_imm(j, k):
        a("global_load_dword v%d, v127, %%[chunks] offset:%d" % (bq(j), 16 * k + 4), "B_%d: Chunk::begin" % k)

    def rec_request(j):
        a("v_lshl_add_u32 v125, v%d, 3, %%[slot8]" % bq(j))
        a("global_load_dwordx2 %s, v125, %%[rb]%s" % (vr(rq(j), 2), rec_policy))

    def row_request(j):
        a("v_lshl_add_u32 v%d, v%d, 8, %%[lds]" % (tq(j), rq(j) + 1))
        a("v_mad_u32_u24 v124, v%d, %%[xrb], %%[lb]" % rq(j))
        a("global_load_dwordx4 %s, v124, %%[xb]" % vr(xq(j), 4))

    def prepare(k):
        """message of chunk k (stage k % D, set k % 4, whose relation row is already there) -> transposed, in registers;
        then the stage is refilled and the relation row of chunk k + 1 is requested"""
        m = SETS[k % 4]
        # chunk k needs G_k, R_k+D and B_k+2D: all three went out when chunk k - D was prepared (or in the prologue in the
        # same order), G first, and every later preparation appended three requests -> 3 (D - 1) requests are younger.
        if not os.environ.get("ULTRA_GEN_PROD_NOWAIT"):   # (measurement builds: what the step costs without the wait; results are wrong)
            a("s_waitcnt vmcnt(%d)" % (3 * (D - 1)))
        x = xq(k)
        a("%s %s, %s, %s" % (binop, vr(m, 2), vr(m, 2), vr(x, 2)))
        a("%s %s, %s, %s" % (binop, vr(m + 2, 2), vr(m + 2, 2), vr(x + 2, 2)))
        a("s_nop 1", "VALU write -> v_permlane*_swap read: 2 wait states")
        a("v_permlane32_swap_b32_e32 v%d, v%d" % (m, m + 2))
        a("v_permlane32_swap_b32_e32 v%d, v%d" % (m + 1, m + 3))
        a("s_nop 1")
        a("v_permlane16_swap_b32_e32 v%d, v%d" % (m, m + 1))
        a("v_permlane16_swap_b32_e32 v%d, v%d" % (m + 2, m + 3))
        row_request(k)                                               # G_k+D
        rec_request(k)                                               # R_k+2D
        a("v_add_u32_e32 v127, 16, v127")
        a("global_load_dword v%d, v127, %%[chunks]" % bq(k), "B_k+3D")
        a("ds_read_b128 %s, v%d" % (vr(SETS[(k + 1) % 4], 4), tq(k + 1)), "relation row of the next chunk")

    a("s_mov_b32 %[i], 0")
    a("s_mov_b32 %[half], 0")
    a("v_mov_b32_e32 v127, 0")
    for j in range(D):
        begin_request_imm(j, j)                                      # B_0 .. B_D-1
    a("s_waitcnt vmcnt(0)")
    for j in range(D):
        rec_request(j)                                               # R_0 .. R_D-1
        begin_request_imm(j, D + j)                                  # B_D .. B_2D-1
    a("s_waitcnt vmcnt(0)")
    for j in range(D):
        row_request(j)                                               # G_j
        rec_request(j)                                               # R_D+j
        begin_request_imm(j, 2 * D + j)                              # B_2D+j      sequence: G_0 R_D B_2D G_1 R_D+1 B_2D+1 ...
    a("v_mov_b32_e32 v127, 0x%x" % (16 * (3 * D - 1) + 4), "descriptor offset of chunk 3 D - 1")
    a("ds_read_b128 %s, v%d" % (vr(SETS[0], 4), tq(0)), "relation row of chunk 0")
    a("s_waitcnt lgkmcnt(0)")
    prepare(0)
    a("s_waitcnt lgkmcnt(0)")
    a.label(".Lprod_loop_%=")
    for J in range(D):
        # step i (i % D = J): park message i -- prepared during the previous step, so the write goes out right behind the
        # barrier and the LDS works on it while message i + 1 is prepared; then barrier i.  (A message past the last chunk
        # is prepared from prefetched, valid data and never parked.)
        # park first, prepare behind it.  (Preparing first -- so that the consumer's reads of the chunk behind the barrier
        # reach the LDS ahead of the fifteen 13-cycle writes -- was measured slower: 735 vs 672 cycles per chunk; the writes
        # then complete later and hold up the barrier.)
        a("v_add_u32_e32 v126, %[half], %[ring]")
        a("ds_write_b128 v126, %s" % vr(SETS[J % 4], 4))
        a("s_xor_b32 %%[half], %%[half], %d" % RING_HALF_BYTES)
        prepare(J + 1)
        a("s_add_i32 %[i], %[i], 1")
        a("s_waitcnt lgkmcnt(0)")
        a("s_barrier")
        a("s_cmp_ge_i32 %[i], %[n]")
        a("s_cbranch_scc1 .Lprod_done_%=")
    a("s_branch .Lprod_loop_%=")
    a.label(".Lprod_done_%=")
    a("s_waitcnt vmcnt(0)", "requests past the last chunk: nobody consumes them")
    return a


def clobbers(lo, hi):
    return ", ".join('"v%d"' % r for r in range(lo, hi + 1))


HEADER = '''// GENERATED by tools/gen_order_asm.py -- do not edit; edit the generator and re-run it.
//
// The two software-pipelined inner loops of rspmm_order_kernel (fp32, unit weights, relation slice in LDS) as gfx950
// assembly: see the generator's docstring for why these are not left to the compiler and for how every wait count
// follows from the issue order.  Arithmetic and summation order are those of the C++ paths (walk_row_in_order and the
// producer lambda in rspmm_order_kernels.hpp), i.e. rspmm.cpp:61-72's.
#pragma once

#include <hip/hip_runtime.h>

namespace ultra {


'''


def main():
    parts = [HEADER]
    parts.append("// group stream of this lane's 16-lane group: rem = the stream's length in steps (edges + one marker per row), roff =\n"
                 "// byte offset of the lane's first record ((begin + lane % 8) * 8), l8 = lane % 8, lb = lane's byte offset inside a row,\n"
                 "// lds = LDS byte address of the lane's part of relation row 0, mark = lds + 256 * num_rel, bndoff = byte offset of the\n"
                 "// boundary row's part of this lane (0xffffffff: none), b = its boundary values, bz = what a row other than the boundary\n"
                 "// row meets at its flush under min / max (0: the boundary tensor's zeros; -+inf: nothing), ns / nf = steps of the wave's\n"
                 "// longest / shortest stream (ns > 0, wave-uniform), xb / rb / ob = source slice, stream records, output slice.\n"
                 "// POST == 2: every flushed row goes to the workgroup's update waves, into the LDS tiles at byte address qtile (control block at\n"
                 "// qctl), and not to memory (stream_park in the generator; whole-span rows only: lb = 16 (lane % 16)).\n"
                 "// ULTRA_STREAM_DIET (generator: DIET): l8 carries lim = len - lane % 8 instead of lane % 8, rem stays the stream's length, rmk\n"
                 "// = the type a marker record holds (num_rel; num_rel * 256 where the records are pre-shifted).  ULTRA_STREAM_PRESHIFT_GEN: the\n"
                 "// POST variants read records (col * 256, type * 256) -- the twelve-walker schedules' format (plan.hpp ULTRA_STREAM_PRESHIFT).\n"
                 "#define ULTRA_STREAM_DIET " + ("1" if DIET else "0") + "\n#define ULTRA_STREAM_PRESHIFT_GEN " + ("1" if DIET else "0") + "\n"
                 "template <int SUM, int MUL, int POST>\n"
                 "__device__ __forceinline__ void order_stream_asm(int rem, uint32_t roff, const int l8, const uint32_t lb, const uint32_t lds,\n"
                 "                                                 const uint32_t mark, const uint32_t bndoff, const float (&b)[4], const float bz,\n"
                 "                                                 const int ns, const int nf, const char *xb, const char *rb, const char *ob,\n"
                 "                                                 const uint32_t xrb, const uint32_t qctl, const uint32_t qtile,\n"
                 "                                                 const uint32_t rmk) {\n"
                 "    int kb, t0, t1, m32, mb;\n    unsigned long long ex, mk;\n    (void)qtile, (void)qctl, (void)m32, (void)mb, (void)rmk;\n"
                 "    const uint32_t rowpitch = " + str(HANDOFF2_ROW_BYTES) + "u;\n    (void)rowpitch;\n")
    first = True
    for post in (0, 2):
        for sum_code in (0, 1, 2):
            for mul_code in (0, 1):
                a = gen_stream(sum_code, mul_code, REC_POLICY, post)
                cond = "SUM == %d && MUL == %d && POST == %d" % (sum_code, mul_code, post)
                parts.append("    %sif constexpr (%s) {\n" % ("" if first else "else ", cond))
                first = False
                parts.append("        asm volatile(\n" + a.render("            ") + "\n")
                parts.append('            : [rem] "+v"(rem), [roff] "+v"(roff), [kb] "=&s"(kb), [t0] "=&s"(t0), [t1] "=&s"(t1), [ex] "=&s"(ex),\n'
                             '              [mk] "=&s"(mk)%s\n' % (', [m32] "=&s"(m32), [mb] "=&s"(mb)' if DIET else ''))
                parts.append('            : [l8] "v"(l8), [lb] "v"(lb), [lds] "v"(lds), [mark] "v"(mark), [bndoff] "v"(bndoff), [b0] "v"(b[0]),\n'
                             '              [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [ns] "s"(ns), [nf] "s"(nf), [xb] "s"(xb), [rb] "s"(rb),\n'
                             '              [ob] "s"(ob), [xrb] "s"(xrb)%s%s%s%s\n' % (', [bz] "v"(bz)' if sum_code else '', ', [qctl] "s"(qctl)' if post else '',
                                                                                         ', [qtile] "s"(qtile), [rowpitch] "s"(rowpitch)' if post == 2 else '',
                                                                                         ', [rmk] "s"(rmk)' if DIET else ''))
                lo, hi = (STREAM_CLOBBER_LO, 126) if post == 2 else (STREAM_CLOBBER_LO, STREAM_CLOBBER_HI)
                parts.append('            : "memory", "vcc", "scc", %s);\n' % clobbers(lo, hi))
                parts.append("    }\n")
    parts.append("}\n\n")

    parts.append("// chain producers: `n` chunks from descriptor `chunks` on (Chunk = 16 B, begin at +4), one s_barrier per chunk;\n"
                 "// slot8 = 8 * slot, ring = LDS byte address of this lane's 16 B in ring half 0 ((quad * 64 + lane) * 16 past the ring)\n"
                 "template <int MUL>\n"
                 "__device__ __forceinline__ void order_produce_asm(const int n, const void *chunks, const uint32_t slot8, const uint32_t lb,\n"
                 "                                                  const uint32_t lds, const uint32_t ring, const char *xb, const char *rb,\n"
                 "                                                  const uint32_t xrb) {\n"
                 "    int i, half;\n")
    first = True
    for mul_code in (0, 1):
        if True:
            a = gen_producer(mul_code, REC_POLICY)
            cond = "MUL == %d" % mul_code
            parts.append("    %sif constexpr (%s) {\n" % ("" if first else "else ", cond))
            first = False
            parts.append("        asm volatile(\n" + a.render("            ") + "\n")
            parts.append('            : [i] "=&s"(i), [half] "=&s"(half)\n'
                         '            : [n] "s"(n), [chunks] "s"(chunks), [slot8] "v"(slot8), [lb] "v"(lb), [lds] "v"(lds), [ring] "v"(ring),\n'
                         '              [xb] "s"(xb), [rb] "s"(rb), [xrb] "s"(xrb)\n'
                         '            : "memory", "scc", %s);\n' % clobbers(PROD_CLOBBER_LO, PROD_CLOBBER_HI))
            parts.append("    }\n")
    parts.append("}\n\n")
    parts.append("}  // namespace ultra\n")
    with open(OUT, "w") as f:
        f.write("".join(parts))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
