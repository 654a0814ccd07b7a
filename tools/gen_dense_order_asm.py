"""Generator of the MEASUREMENT build of the relation-graph layer's chain (dense_order_layer.hip, phase 1, -DULTRA_DOL_ASM=1) as
gfx950 assembly.  Measured in round 4: bit-exact and not faster than the C++ loop (21.7 vs 21.5 us per layer), so the header is
no longer part of the source tree; it is generated where a measurement build wants it:

    python tools/gen_dense_order_asm.py          # writes ultra_amd/lib/variants/dense_order_asm.hpp (git-ignored)
    python tools/build_variant.py dolasm -DULTRA_DOL_ASM=1 '-DULTRA_DOL_ASM_HEADER="<that path>"'

The chain is one dependent v_mfma_f32_16x16x4_f32 per source column (474 at FB15k237): the wave has nothing else to hide a
memory round trip behind, so its operands -- per 16-column stage one 16-byte adjacency word and sixteen x values per lane -- must
be in registers before their instruction's turn.  hipcc (ROCm 7.2) hoists the products `rel * x` of ALL stages of a loop
iteration to its top and therefore waits for the youngest load of the iteration there (`s_waitcnt vmcnt(0)` a third into the
body, read off the ISA): the queue drains once per 48 columns and the chain runs at half the rate of the instruction
(15 us against 7-8 us per layer).  Here: NS stages in flight, every wait counted from the issue order --

  stage set s (registers X[s][0..15], A[s][0..3]) is refilled for stage j + NS as soon as stage j has consumed it: the x value of
  column q right behind its product, the adjacency word behind the stage's last conversion.  Loads return in order, and every
  stage issues exactly 17, so when stage j's turn comes `vmcnt(17 (NS - 1))` means "stage j has landed".
  The next column's operands (byte -> float, product) are prepared in the shadow of the current matrix instruction.

Arithmetic as in the C++ loop: a = (float) byte, b = fl(rel * x) rounded on its own (rspmm.cpp:67), acc = fma chain over the four
types of a column, columns ascending -- the reference's order (see dense_order_layer.hip).
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("ULTRA_GEN_DENSE_OUT") or os.path.join(os.path.dirname(HERE), "ultra_amd", "lib", "variants", "dense_order_asm.hpp")

NS = int(os.environ.get("ULTRA_GEN_DOL_NS", "4"))         # stages in flight (measurement builds: other depths)
TOUCH = os.environ.get("ULTRA_GEN_DOL_TOUCH", "1") == "1"  # (measurement builds: without the touch loads)
X0 = (256 - (20 * NS + 41)) & ~1   # X[s][q]: 16 NS registers (even: the adjacency words behind them are 4-register tuples)
A0 = X0 + 16 * NS           # A[s][0..3]
TA = A0 + 4 * NS            # a / b operands, two of each (ping-pong)
TB = TA + 2
VADJ = TB + 2               # byte offset of this lane's adjacency word of the stage being fetched
VO0 = VADJ + 1              # 16 registers: q * row_bytes + lane_bytes (regular stages, on top of the stage offset)
VL0 = VO0 + 16              # 16 registers: the clamped offsets of the graph's last stage
VT = VL0 + 16               # 2 registers: address of the load being issued (ping-pong)
VDUMMY, VTOUCH = VT + 2, VT + 3   # destination of the touch loads (never read), their offset
CLOBBER_LO, CLOBBER_HI = X0, VTOUCH
assert CLOBBER_HI <= 255 and X0 >= 64 and A0 % 2 == 0


def x(s, q):
    return X0 + 16 * s + q


def areg(s, w):
    return A0 + 4 * s + w


class Asm:
    def __init__(self):
        self.lines = []
        self.n = 0

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


def fetch_x(a, s, q, last):
    """x value of column q of the stage being fetched into set s.  Regular stage: the lane's loop-invariant offset + the stage
    offset (a scalar, advanced per stage); the graph's last stage (and the dummy stages behind it): clamped per-lane offsets."""
    if last:
        a("global_load_dword v%d, v%d, %%[xb]" % (x(s, q), VL0 + q))
    else:
        t = VT + (q & 1)
        a("v_add_u32_e32 v%d, %%[sstage], v%d" % (t, VO0 + q))
        a("global_load_dword v%d, v%d, %%[xb]" % (x(s, q), t))


def fetch_adj(a, s):
    a("global_load_dwordx4 v[%d:%d], v%d, %%[ap]" % (areg(s, 0), areg(s, 3), VADJ))


def advance_fetch(a):
    """the fetch cursor moves on by one stage (stays on the graph's last stage once it is there)"""
    a("s_add_i32 %[jf], %[jf], 1")
    a("s_cmp_le_i32 %[jf], %[nlast]")
    a("s_cbranch_scc0 .Ldol_stay%d_%%=" % a.n)
    a("s_add_u32 %[sstage], %[sstage], %[stagebytes]")
    a("v_add_u32_e32 v%d, 0x400, v%d" % (VADJ, VADJ))
    a.label(".Ldol_stay%d_%%=" % a.n)
    a.n += 1


def gen_chain():
    a = Asm()
    # ---- per-lane offsets ----
    a("v_mov_b32_e32 v%d, %%[lb]" % VO0)
    for q in range(1, 16):
        a("v_add_u32_e32 v%d, %%[rowbytes], v%d" % (VO0 + q, VO0 + q - 1))
    for q in range(16):
        a("s_lshl_b32 %[t0], %[nlast], 4")
        if q:
            a("s_add_i32 %%[t0], %%[t0], %d" % q)
        a("s_min_i32 %[t0], %[t0], %[nin1]", "a column past the graph reads the last row (its adjacency bytes are 0)")
        a("s_mul_i32 %[t0], %[t0], %[rowbytes]")
        a("v_add_u32_e32 v%d, %%[t0], %%[lb]" % (VL0 + q))
    # ---- touch: the sample's x slice and this tile's adjacency words, one 128-byte line per lane and request ----
    # The layer's input was written by the previous launch from all eight XCDs, so it is in no L2 when this launch starts, and
    # every workgroup of a sample walks its rows at the same pace: without this each stage's first reader pays a trip to the
    # Infinity Cache that no lookahead of a few stages covers (measured: the chain runs at 67 cycles per instruction instead
    # of 36, whatever the depth of the register queue).  Fire and forget: the requests are the oldest of the wave's queue, the
    # first counted wait below collects them -- one trip for the whole slice instead of one per stage.
    a("v_mov_b32_e32 v%d, %%[touch0]" % VTOUCH)
    a("s_mov_b32 %[t0], 0")
    if not TOUCH:
        a("s_branch .Ldol_touched_%=")
    a.label(".Ldol_touch_%=")
    a("v_min_u32_e32 v%d, %%[xlast], v%d" % (VT, VTOUCH))
    a("global_load_dword v%d, v%d, %%[xb]" % (VDUMMY, VT))
    a("v_add_u32_e32 v%d, 0x8000, v%d" % (VTOUCH, VTOUCH), "four waves x 64 lanes x 128 B")
    a("s_add_i32 %[t0], %[t0], 1")
    a("s_cmp_lt_i32 %[t0], %[ntouch]")
    a("s_cbranch_scc1 .Ldol_touch_%=")
    a("v_min_u32_e32 v%d, %%[alast], %%[touch0]" % VT)
    a("global_load_dword v%d, v%d, %%[ap]" % (VDUMMY, VT), "(a tile's adjacency words: 1 KB per stage, 256 lines cover 32 stages; longer graphs touch their head only)")
    a.label(".Ldol_touched_%=")
    # ---- prologue: stages 0 .. NS - 1 ----
    a("s_mov_b32 %[jf], 0")
    a("s_mov_b32 %[sstage], 0")
    a("v_mov_b32_e32 v%d, %%[adj0]" % VADJ)
    for s in range(NS):
        # (regular or last: decided per stage -- a graph of fewer than NS stages has its last stage early)
        a("s_cmp_lt_i32 %[jf], %[nlast]")
        a("s_cbranch_scc0 .Ldol_pro_last%d_%%=" % s)
        for q in range(16):
            fetch_x(a, s, q, False)
        a("s_branch .Ldol_pro_done%d_%%=" % s)
        a.label(".Ldol_pro_last%d_%%=" % s)
        for q in range(16):
            fetch_x(a, s, q, True)
        a.label(".Ldol_pro_done%d_%%=" % s)
        fetch_adj(a, s)
        advance_fetch(a)
    a("s_mov_b32 %[j], 0")
    a.label(".Ldol_loop_%=")
    for s in range(NS):
        a("s_waitcnt vmcnt(%d)" % (17 * (NS - 1)), "stage j has landed: the %d younger requests belong to the other sets" % (17 * (NS - 1)))
        # refill variant of this stage (what is fetched into set s while it is consumed): regular or last
        a("s_cmp_lt_i32 %[jf], %[nlast]")
        a("s_cbranch_scc0 .Ldol_body_last%d_%%=" % s)
        for last in (False, True):
            if last:
                a.label(".Ldol_body_last%d_%%=" % s)
            # operands of column 0
            a("v_cvt_f32_ubyte0_e32 v%d, v%d" % (TA, areg(s, 0)))
            a("v_mul_f32_e32 v%d, %%[relv], v%d" % (TB, x(s, 0)))
            fetch_x(a, s, 0, last)
            for q in range(16):
                cur = q & 1
                nxt = cur ^ 1
                if q + 1 < 16:
                    w, byte = (q + 1) >> 2, (q + 1) & 3
                    a("v_cvt_f32_ubyte%d_e32 v%d, v%d" % (byte, TA + nxt, areg(s, w)))
                    a("v_mul_f32_e32 v%d, %%[relv], v%d" % (TB + nxt, x(s, q + 1)))
                a("v_mfma_f32_16x16x4_f32 %%[acc], v%d, v%d, %%[acc]" % (TA + cur, TB + cur))
                if q + 1 < 16:
                    fetch_x(a, s, q + 1, last)      # its product has been taken: the register is free for stage j + NS
            fetch_adj(a, s)                         # (behind the stage's last conversion)
            if not last:
                a("s_branch .Ldol_body_done%d_%%=" % s)
        a.label(".Ldol_body_done%d_%%=" % s)
        advance_fetch(a)
        a("s_add_i32 %[j], %[j], 1")
        a("s_cmp_ge_i32 %[j], %[njc]")
        a("s_cbranch_scc1 .Ldol_end_%=")
    a("s_branch .Ldol_loop_%=")
    a.label(".Ldol_end_%=")
    a("s_waitcnt vmcnt(0)", "the refills past the last stage: nobody consumes them")
    return a


HEADER = '''// GENERATED by tools/gen_dense_order_asm.py -- do not edit; edit the generator and re-run it.
//
// The chain of dependent matrix instructions of the relation-graph layer (dense_order_layer.hip, phase 1) as gfx950 assembly:
// see the generator's docstring for why, and for how every wait count follows from the issue order.
#pragma once

#include <hip/hip_runtime.h>

namespace ultra {

using dol_f32x4 = float __attribute__((ext_vector_type(4)));

// acc: the wave's 16 x 16 accumulator tile; njc stages of 16 source columns over a graph of n_in columns; adj0 = byte offset of
// this lane's adjacency word of stage 0 behind `ap` (64 lanes x 16 B per stage); xb = the sample's x slice (rows of row_bytes
// bytes), lane_bytes = byte offset of this lane's column inside a row; relv = this lane's relation value.
// touch0 = 128 * (this thread's index in the workgroup of four waves): the line it touches first.
__device__ __forceinline__ void dense_order_chain_asm(dol_f32x4 &acc, const int njc, const int n_in, const uint32_t adj0, const void *ap,
                                                      const char *xb, const uint32_t lane_bytes, const uint32_t row_bytes, const float relv,
                                                      const uint32_t touch0) {
    int j, jf, t0;
    uint32_t sstage;
    const int nlast = njc - 1, nin1 = n_in - 1;
    const uint32_t stagebytes = 16u * row_bytes;
    const uint32_t xlast = (uint32_t)n_in * row_bytes - 4u, alast = (uint32_t)njc * 1024u - 4u;
    const int ntouch = (int)(((uint32_t)n_in * row_bytes + 0x7fffu) >> 15);
'''


def main():
    a = gen_chain()
    parts = [HEADER]
    parts.append("    asm volatile(\n" + a.render("        ") + "\n")
    ops_out = '[acc] "+v"(acc), [j] "=&s"(j), [jf] "=&s"(jf), [t0] "=&s"(t0), [sstage] "=&s"(sstage)'
    ins = ['[njc] "s"(njc)', '[nlast] "s"(nlast)', '[nin1] "s"(nin1)', '[adj0] "v"(adj0)', '[ap] "s"(ap)', '[xb] "s"(xb)',
           '[relv] "v"(relv)', '[stagebytes] "s"(stagebytes)', '[lb] "v"(lane_bytes)', '[rowbytes] "s"(row_bytes)',
           '[touch0] "v"(touch0)', '[xlast] "s"(xlast)', '[alast] "s"(alast)', '[ntouch] "s"(ntouch)']
    clob = ", ".join('"v%d"' % r for r in range(CLOBBER_LO, CLOBBER_HI + 1))
    parts.append("        : %s\n        : %s\n        : \"memory\", \"scc\", %s);\n}\n\n}  // namespace ultra\n" % (ops_out, ", ".join(ins), clob))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write("".join(parts))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
