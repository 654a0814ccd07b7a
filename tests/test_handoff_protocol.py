"""The LDS hand-off of rspmm_order_kernel's form 3 (walkers park rows, update waves take 32-row blocks) as an
executable model: the control words and the decisions of csrc/rspmm_order_kernels.hpp (update waves) and
tools/gen_order_asm.py stream_park (walkers), one Python generator per wave / group, an adversarial random scheduler in place of
the hardware's.  Every LDS operation is one scheduling point; LDS operations of ONE wave take effect in program order, as on
the chip.  Checked on every interleaving tried: no ring row is overwritten before all four update waves are done with it, an
update wave never multiplies a row that is not the generation it believes it is, pre-norm rows replace x rows only once
everybody has read them, every parked row is finished exactly once, and everybody terminates (row counts 0 .. 200: empty,
partial and full last blocks -- the round-4 hang was a partial last tile whose row count equalled the "not complete yet" value
of the look loop; the row counts are clamped to [0, 16] since).  No GPU needed; three sensitivity tests break one rule each and
expect the model to notice."""
import random

import pytest

NT = 4                  # ring tiles of 16 rows (UPD2_NT)
WALKERS = 12            # walker waves, four 16-lane groups each
UPDATERS = 4


class Lds(object):
    def __init__(self):
        self.tail = 0                   # ctl[0]
        self.walked = 0                 # ctl[1]
        self.meeting = 0                # ctl[2]: arrivals of the update waves
        self.consumed = 0               # ctl[4]: update waves counted out of blocks (four per block of two generations)
        self.posted = [0] * NT          # rows posted into ring tile b (monotonic)
        self.rowid = [None] * (16 * NT)
        self.agg = [None] * (16 * NT)   # ring rows: (generation, row) tags
        self.x = [None] * (16 * NT)     # x rows; a block's pre-norm rows take their place: ("y", generation, row)
        self.readers_done = {}          # generation -> update waves done with it (the model's own bookkeeping)


FAULT = {"name": None}      # the sensitivity tests break one rule of the protocol at a time


def walker_group(lds, rows, wave_left, errors):
    """One 16-lane group of a walker wave: flush after flush (stream_park)."""
    for row in rows:
        slot = lds.tail                 # ds_add_rtn tail (with the read of `consumed`: one round trip)
        lds.tail += 1
        seen = lds.consumed
        yield
        gen = slot // 16
        while FAULT["name"] != "no backpressure" and not (2 * (seen >> 2) + NT > gen):   # the buffer is free once generation gen - NT has been counted out by all four
            seen = lds.consumed
            yield
        pos = slot % (16 * NT)
        old = lds.agg[pos]
        if old is not None and lds.readers_done.get(old[0], 0) < UPDATERS:
            errors.append("generation %d row overwritten by generation %d before every update wave was done with it" % (old[0], gen))
        lds.agg[pos] = (gen, row)       # ds_write_b128 (aggregate), ds_write_b128 (x row), ds_write_b32 (row id): in order
        lds.x[pos] = (gen, row)
        lds.rowid[pos] = row
        yield
        lds.posted[gen % NT] += 1       # ds_add posted: behind the writes
        yield
    wave_left[0] -= 1
    if wave_left[0] == 0:               # the wave's last group: walked++ (behind everything the wave did)
        lds.walked += 1
    yield


def update_wave(lds, u, finished, errors, not_yet):
    """Update wave u (rspmm_order_kernels.hpp, UPDATE == 3; no chain blocks in the model)."""
    epoch = 0

    def look(B, wait):
        b0 = (2 * B) % NT
        need = 16 * (B // (NT // 2) + 1)
        while True:
            walked, have0, have1, tail = lds.walked, lds.posted[b0], lds.posted[b0 + 1], lds.tail
            yield
            if have0 >= need and have1 >= need:
                return 16, 16
            if walked == WALKERS:
                rem = tail - 32 * B
                return max(0, min(16, rem)), max(0, min(16, rem - 16))
            if not wait:
                return not_yet, not_yet
            yield

    ahead, nxt = False, None
    t = 0
    while True:
        if ahead:
            n0, n1 = nxt
        else:
            n0, n1 = yield from look(t, True)
            if n0 <= 0:
                break
        b0 = (2 * t) % NT
        rows = [(2 * t, r) for r in range(n0)] + [(2 * t + 1, r) for r in range(n1)]
        # operand reads of the x rows and of the aggregate rows
        for gen, r in rows:
            pos = (gen % NT) * 16 + r
            if lds.x[pos] is None or lds.x[pos][0] != gen or lds.agg[pos] is None or lds.agg[pos][0] != gen:
                errors.append("update wave %d block %d: ring row %d holds %r / %r, not generation %d" % (u, t, pos, lds.x[pos], lds.agg[pos], gen))
        yield
        epoch += UPDATERS
        lds.meeting += 1                # arrive: this wave's x reads are served
        yield
        while FAULT["name"] != "no first meeting" and lds.meeting < epoch:      # meet: everybody has read the block's x rows
            yield
        for gen, r in rows:             # the pre-norm rows take the x rows' place (this wave's 16 features of every row)
            pos = (gen % NT) * 16 + r
            tag = lds.x[pos]
            if tag[0] != "y":
                lds.x[pos] = ("y", gen, tag[1], 1)
            else:
                lds.x[pos] = ("y", gen, tag[2], tag[3] + 1)
        yield
        epoch += UPDATERS
        lds.meeting += 1
        yield
        # while the others arrive: is the next block complete already?  (kept privately: n0, n1 of what THIS wave saw)
        p0, p1 = yield from look(t + 1, False)
        ahead = p0 > 0
        nxt = (p0, p1)
        while lds.meeting < epoch:
            yield
        # finishing: this wave's rows rr and 16 + rr of the block need all four waves' features
        for gen, r in rows:
            if r % UPDATERS != u:
                continue
            pos = (gen % NT) * 16 + r
            tag = lds.x[pos]
            if tag[0] != "y" or tag[1] != gen or tag[3] != UPDATERS:
                errors.append("update wave %d finishes generation %d row %d from %r" % (u, gen, r, tag))
            finished.append(lds.rowid[pos])
        yield
        for gen in (2 * t, 2 * t + 1):
            lds.readers_done[gen] = lds.readers_done.get(gen, 0) + 1
        lds.consumed += 4 if (FAULT["name"] == "first wave releases the block" and u == 0) else (0 if FAULT["name"] == "first wave releases the block" else 1)
        yield                           # (counted out of the block: at four the walkers may reuse its rows)
        t += 1


def run(seed, n_rows, not_yet=-64, bias=None):
    rng = random.Random(seed)
    lds = Lds()
    errors, finished = [], []
    rows = list(range(n_rows))
    rng.shuffle(rows)
    groups = [[] for _ in range(4 * WALKERS)]
    for r in rows:
        groups[rng.randrange(len(groups))].append(r)
    tasks = []
    for w in range(WALKERS):
        left = [4]
        for g in range(4):
            tasks.append(walker_group(lds, groups[4 * w + g], left, errors))
    for u in range(UPDATERS):
        tasks.append(update_wave(lds, u, finished, errors, not_yet))
    weights = [1.0] * len(tasks)
    if bias == "slow updaters":
        weights[-UPDATERS:] = [0.05] * UPDATERS
    elif bias == "slow walkers":
        weights[:4 * WALKERS] = [0.05] * (4 * WALKERS)
    elif bias == "one lazy updater":
        weights[-1] = 0.02
    alive = list(range(len(tasks)))
    steps = 0
    while alive:
        steps += 1
        if steps > 4_000_000:
            return errors + ["no termination (%d tasks left, tail %d, walked %d, consumed %d)" % (len(alive), lds.tail, lds.walked, lds.consumed)], finished
        k = rng.choices(alive, weights=[weights[i] for i in alive])[0]
        try:
            next(tasks[k])
        except StopIteration:
            alive.remove(k)
    return errors, finished


@pytest.mark.parametrize("bias", [None, "slow updaters", "slow walkers", "one lazy updater"])
@pytest.mark.parametrize("n_rows", [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 64, 65, 200])
def test_every_parked_row_is_finished_once_whatever_the_interleaving(n_rows, bias):
    for seed in range(6):
        errors, finished = run(seed, n_rows, bias=bias)
        assert not errors, errors[:3]
        assert sorted(finished) == list(range(n_rows))


@pytest.mark.parametrize("fault,bias", [("no backpressure", "slow updaters"), ("no first meeting", "one lazy updater"),
                                        ("first wave releases the block", "one lazy updater")])
def test_the_model_notices_a_broken_rule(fault, bias):
    """Sensitivity: without the walkers' wait for a free buffer, without the meeting between reading the x rows and writing
    the pre-norm rows over them, or with ONE update wave releasing a block for all four, some interleaving goes wrong."""
    FAULT["name"] = fault
    try:
        found = False
        for seed in range(12):
            errors, finished = run(seed, 200, bias=bias)
            if errors or sorted(finished) != list(range(200)):
                found = True
                break
        assert found
    finally:
        FAULT["name"] = None
