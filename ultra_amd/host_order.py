"""Summation order of the readout's last product, taken from the host BLAS.

EntityNBFNet's readout ends in nn.Linear(feature_dim, 1) (models.py:208): one dot product of 128 terms per candidate.
On the reference's CPU path that is an MKL GEMV whose association of the 128 products depends on the machine -- 16-lane
AVX-512 accumulators with a masked tail on an Intel Xeon, four strided SSE-style chains on an AMD EPYC (both observed with
the same torch build).  Scores only reproduce the reference bit for bit -- and rankings at near-ties -- if the GPU adds the
products in the same association, so the readout kernel takes the association as DATA (a small program) and this module
recovers it from the BLAS the process itself links:

  probe_tree(K)          the summation tree of F.linear(x, w) for w of shape (1, K), by probing: a row with x_i = 1,
                         x_j = -1, x_k = 2^-30 (all else 0, w = 1) gives 2^-30 iff i and j are added before k joins them
                         ((1 + 2^-30) rounds to 1 in fp32).  O(K^2) probe rows per tree level, ~1 s for K = 128.
  tree_to_stages(tree)   the tree in the form the kernel executes: a sequence of STAGES; a stage has L lanes (L a power of
                         two), lane p runs a chain  acc = fma(h[k], w[k], acc)  (or acc + fl(h[k] w[k]) where the host
                         code does not fuse) over its element list starting from 0 (lane 0 of a later stage: from the
                         previous stage's result), then the lanes are folded
                         v[p] += v[p + L/2], v[p] += v[p + L/4], ...  Raises ValueError for trees outside this family.
  emulate(stages, x, w)  numpy restatement of what the kernel does (tests; validation against torch on this host).
  readout_program(K)     the validated program of this host as an int32 list (cached), or the k-ascending single chain
                         when the host's tree is outside the family or ULTRA_READOUT_ORDER=sequential.

Only the association is taken from the host; every product and sum is an fp32 operation on the GPU.
"""
import os
import subprocess
import warnings

import numpy as np
import torch

TINY = 2.0 ** -30
PROBE_ROWS = 8192      # rows per probing call
PROBE_SPARE = 64       # ... of which the last ones stay empty: a BLAS sums the trailing rows of a call with a remainder kernel


class _single_thread(object):
    """Probing calls run on one thread: with several, every thread's share of the rows ends in remainder rows that follow
    another association, and probes landing there would contradict the others."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *exc):
        torch.set_num_threads(self.n)



def _run_queries(qs, K):
    out = []
    w = torch.ones(1, K)
    cap = PROBE_ROWS - PROBE_SPARE
    with _single_thread():
        for s in range(0, len(qs), cap):
            chunk = qs[s:s + cap]
            x = torch.zeros(PROBE_ROWS, K)
            idx = torch.tensor(chunk, dtype=torch.long)
            r = torch.arange(len(chunk))
            x[r, idx[:, 0]] = 1.0
            x[r, idx[:, 1]] = -1.0
            x[r, idx[:, 2]] = TINY
            y = torch.nn.functional.linear(x, w)[:, 0]
            out += (y[:len(chunk)] != 0).tolist()
    return out


def probe_tree(K=128):
    """Nested pairs over 0..K-1: (a, b) = a and b are added; leaves are products h[k] * w[k]."""

    def solve(leaves):
        if len(leaves) == 1:
            return leaves[0]
        if len(leaves) == 2:
            return (leaves[0], leaves[1])
        a, rest = leaves[0], leaves[1:]
        qs = [(a, j, k) for j in rest for k in rest if j != k]
        res = iter(_run_queries(qs, K))
        below = {(j, k): next(res) for j in rest for k in rest if j != k}     # LCA(a, j) strictly below LCA(a, k)
        level = {j: sum(1 for k in rest if k != j and below[(k, j)]) for j in rest}
        groups = {}
        for j in rest:
            groups.setdefault(level[j], []).append(j)
        node = a
        for lv in sorted(groups):
            node = (node, solve(groups[lv]))
        return node

    return solve(list(range(K)))


def _as_tuple(t):
    return t if isinstance(t, int) else (_as_tuple(t[0]), _as_tuple(t[1]))


def _leaves(t):
    return [t] if isinstance(t, int) else _leaves(t[0]) + _leaves(t[1])


def annotate(tree, K=128):
    """The probed tree with every join that involves a single product marked as fused or not:
    ('f', acc, k): fma(h[k], w[k], acc) -- the product of leaf k is not rounded on its own; ('a', left, right): an fp32
    add of two finished values.  Probe: w = 1 + 2^-12 everywhere, x_a = -(1 + 2^-12), x_b = 1 + 2^-12: the products are
    -+(1 + 2^-11 + 2^-24), which round to -+(1 + 2^-11); the row sums to +2^-24 if b is fused onto a value holding a's
    rounded product, -2^-24 the other way round, 0 if both products are rounded before they are added."""
    tree = _as_tuple(tree)
    nodes = []

    def collect(t):
        if isinstance(t, int):
            return
        a, b = t
        if isinstance(a, int) or isinstance(b, int):
            nodes.append(t)
        collect(a)
        collect(b)

    collect(tree)
    w = torch.full((1, K), 1.0 + 2.0 ** -12)
    x = torch.zeros(max(len(nodes), 1), K)
    for r, (a, b) in enumerate(nodes):
        if isinstance(b, int):
            acc_leaf, leaf = _leaves(a)[0], b
        else:
            acc_leaf, leaf = _leaves(b)[0], a
        x[r, acc_leaf] = -(1.0 + 2.0 ** -12)
        x[r, leaf] = 1.0 + 2.0 ** -12
    pad = torch.zeros(max(PROBE_ROWS - x.shape[0], PROBE_SPARE), K)
    with _single_thread():
        y = torch.nn.functional.linear(torch.cat([x, pad]), w)[:len(nodes), 0].tolist()
    verdict = {id(n): v for n, v in zip(nodes, y)}

    def build(t):
        if isinstance(t, int):
            return t
        a, b = t
        v = verdict.get(id(t), 0.0)
        if isinstance(b, int) and v > 0:
            return ("f", build(a), b)
        if isinstance(b, int) and v < 0 and isinstance(a, int):
            return ("f", b, a)
        if isinstance(a, int) and v > 0:
            return ("f", build(b), a)
        if isinstance(a, int) and v < 0 and isinstance(b, int):
            return ("f", a, b)
        return ("a", build(a), build(b))

    return build(tree)


UNFUSED = 256      # flag on an element of a lane: v = v + fl(h[k] * w[k]) instead of v = fma(h[k], w[k], v)


def _is_fold(node):
    return not isinstance(node, int) and node[0] == "a" and not isinstance(node[1], int) and not isinstance(node[2], int)


def tree_to_stages(annotated):
    """[(L, carry, [lane element lists])] from the first stage to the last, for an annotate()d tree.  An element is the
    product index k, + UNFUSED where the product is rounded before it is added.  Adding a single rounded product to a lane
    is the same arithmetic whether one calls it a chain step or a fold with a one-element lane; it is parsed as a step."""
    stages = []
    t = annotated
    while t is not None:
        lanes, depth = {}, [0]

        def walk(node, level, lane):
            if _is_fold(node):
                # addition commutes: the operand holding the smallest product index (the side a carried value lives on)
                # keeps the lane, the other one sits `1 << level` lanes away
                first, second = sorted((node[1], node[2]), key=lambda c: min(_leaves_annotated(c)))
                walk(first, level + 1, lane)
                walk(second, level + 1, lane | (1 << level))
                return
            depth[0] = max(depth[0], level)
            lanes[lane] = node

        walk(t, 0, 0)
        L = 1 << depth[0]
        if L > 16:
            raise ValueError("more than 16 lanes")
        carry, lists = None, []
        for p in range(L):
            elems, node = [], lanes.get(p)
            while node is not None:
                if isinstance(node, int):
                    elems.append(node)              # first product of a lane: fma onto +0 = the rounded product
                    node = None
                elif node[0] == "f":
                    elems.append(node[2])
                    node = node[1]
                elif isinstance(node[2], int):
                    elems.append(node[2] + UNFUSED)
                    node = node[1]
                elif isinstance(node[1], int):
                    elems.append(node[1] + UNFUSED)
                    node = node[2]
                else:                               # a fold below a chain: the value carried in from the previous stage
                    if p != 0 or carry is not None:
                        raise ValueError("a carried value outside lane 0")
                    carry, node = node, None
            lists.append(elems[::-1])
        stages.append((L, carry is not None, lists))
        t = carry
    stages.reverse()
    covered = sorted(k & 255 for _, _, lists in stages for lane in lists for k in lane)
    if stages[0][1] or covered != sorted(_leaves_annotated(annotated)):
        raise ValueError("stages do not cover every product exactly once")
    return stages


def _leaves_annotated(t):
    return [t] if isinstance(t, int) else _leaves_annotated(t[1]) + _leaves_annotated(t[2])


PAD = 128          # padding element of the kernel's program: h[128] = w[128] = 0, fma(0, 0, v) = v


def stages_to_program(stages):
    """int32 words for the kernel:
        [n_stage] then one header per stage: [L, carry, (offset_p, groups_p) for p < L], then the element area.
    offset_p: word index (a multiple of 4) of lane p's elements; groups_p: its element count in groups of eight, the
    last group padded with PAD (the kernel reads eight elements at a time, lanes of a stage run in parallel)."""
    header_words = 1 + sum(2 + 2 * L for L, _, _ in stages)
    base = (header_words + 3) // 4 * 4
    prog, area = [len(stages)], []
    for L, carry, lists in stages:
        prog += [L, 1 if carry else 0]
        for lane in lists:
            groups = (len(lane) + 7) // 8
            prog += [base + len(area), groups]
            area += list(lane) + [PAD] * (8 * groups - len(lane))
    return prog + [0] * (base - len(prog)) + area


def sequential_stages(K=128):
    return [(1, False, [list(range(K))])]


def emulate(stages, x, w):
    """What the readout kernel computes for y = x @ w (x (M, K), w (K,)): fp32, the fma chains and folds of `stages`."""
    x64, w64 = np.asarray(x, dtype=np.float64), np.asarray(w, dtype=np.float64)

    def r32(a):
        return a.astype(np.float32).astype(np.float64)

    s = np.zeros(x64.shape[0])
    for L, carry, lists in stages:
        v = []
        for p, elems in enumerate(lists):
            acc = s if (p == 0 and carry) else np.zeros(x64.shape[0])
            for e in elems:
                k = e & 255
                if e & UNFUSED:
                    acc = r32(r32(x64[:, k] * w64[k]) + acc)
                else:
                    acc = r32(x64[:, k] * w64[k] + acc)       # fp32 fma: the 48-bit product is exact in fp64
            v.append(acc)
        half = L >> 1
        while half >= 1:
            for p in range(half):
                v[p] = r32(v[p] + v[p + half])
            half >>= 1
        s = v[0]
    return s.astype(np.float32)


_CACHE = {}
PROGRAM_MAX_WORDS = 640    # READOUT_ORDER_MAX of the readout kernel (csrc/dense_kernels.hip)


def order_id(stages):
    """Short name of a summation order: 'order-' + 8 hex digits of the SHA-1 of its kernel program.  Two runs with the same
    id add the readout's 128 products in the same association (printed in bench.py's config and parity block)."""
    import hashlib
    words = np.asarray(stages_to_program(stages), dtype=np.int32)
    return "order-" + hashlib.sha1(words.tobytes()).hexdigest()[:8]


def _host_key(K):
    """What the association depends on: the CPU model, the torch build and its BLAS."""
    import hashlib
    import platform
    cpu = platform.processor() or platform.machine()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    blas = ""
    try:
        blas = torch.__config__.show()
        blas = " ".join(l.strip() for l in blas.splitlines() if "MKL" in l or "BLAS" in l or "LAPACK" in l)
    except Exception:
        pass
    text = "%s|torch %s|%s|K=%d" % (cpu, torch.__version__, blas, K)
    return hashlib.sha1(text.encode()).hexdigest()[:16], text


def _cache_dir():
    d = os.environ.get("ULTRA_ORDER_CACHE_DIR")
    if not d:
        d = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "ultra_amd")
    return d


def save_stages(path, stages, source, K=128, host=""):
    import json
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump({"format": 1, "K": K, "id": order_id(stages), "source": source, "host": host,
                   "stages": [[L, bool(carry), [list(map(int, lane)) for lane in lists]] for L, carry, lists in stages]}, f)
    os.replace(tmp, path)      # atomic: ranks of one node may write the same file


def load_stages(path, K=128):
    """(stages, source) of an order file written by save_stages / `python -m ultra_amd.host_order --save FILE`: run host B
    with host A's association (ULTRA_READOUT_ORDER=FILE) to reproduce A's score bits."""
    import json
    with open(path) as f:
        rec = json.load(f)
    if rec.get("format") != 1 or rec.get("K") != K:
        raise ValueError("%s is not a readout order file for K = %d" % (path, K))
    stages = [(int(L), bool(carry), [list(map(int, lane)) for lane in lists]) for L, carry, lists in rec["stages"]]
    _check_stages(stages, K)
    return stages, rec.get("source", "file")


def _check_stages(stages, K):
    covered = sorted(e & 255 for _, _, lists in stages for lane in lists for e in lane)
    if covered != list(range(K)):
        raise ValueError("stages do not cover every product exactly once")
    if len(stages_to_program(stages)) > PROGRAM_MAX_WORDS:
        raise ValueError("summation program of %d words exceeds the readout kernel's %d"
                         % (len(stages_to_program(stages)), PROGRAM_MAX_WORDS))


def probe_host(K=128):
    """(stages, source) probed from this process's BLAS and validated against torch on random rows; raises ValueError
    when the tree is outside the family, the program too long, or the emulation misses torch.  Toggles the process-wide
    torch thread count while it runs (readout_stages runs it in a helper process for that reason)."""
    cand = tree_to_stages(annotate(probe_tree(K), K))
    _check_stages(cand, K)
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(4096, K, generator=g), torch.randn(1, K, generator=g)
    with _single_thread():          # one share: at most a handful of remainder rows among the 4096
        want = torch.nn.functional.linear(x, w)[:, 0].numpy()
    match = float((emulate(cand, x.numpy(), w[0].numpy()) == want).mean())
    if match < 0.99:                # (the BLAS sums a few trailing rows of each thread's share with another kernel)
        raise ValueError("probed GEMV order reproduces only %.1f %% of rows" % (100 * match))
    return cand, "host BLAS (probed; %.2f %% of 4096 random rows bit-equal)" % (100 * match)


def _match_fraction(stages, K=128, rows=4096):
    """Share of random rows on which the emulated association equals this process's F.linear(x, (1, K)) bit for bit."""
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(rows, K, generator=g), torch.randn(1, K, generator=g)
    want = torch.nn.functional.linear(x, w)[:, 0].numpy()
    return float((emulate(stages, x.numpy(), w[0].numpy()) == want).mean())


def adopt(stages, source, K=128):
    """Use this association in this process from now on (a multi-GPU job: every rank adds in rank 0's association --
    distributed.share_readout_order)."""
    _check_stages(stages, K)
    _CACHE[K] = (stages, source)
    from . import dense
    # device copies of the previous program: RETIRED, not freed -- a hipGraph captured before this call may still read them
    # (share_readout_order is documented to run before the first forward; a dangling pointer must not be the price of calling it late)
    dense._ORDER_RETIRED.extend(dense._ORDER_CACHE.values())
    dense._ORDER_CACHE.clear()


def _probe_in_subprocess(K, path):
    """The probe in a helper process: O(K^2) single-threaded GEMV calls and torch.set_num_threads(1) stay out of the
    caller's process (data-loader or OpenMP threads there keep their thread count).  The helper links the same torch, i.e.
    the same BLAS.  Returns (stages, source) read back from `path`."""
    import subprocess
    import sys
    env = dict(os.environ, ULTRA_READOUT_ORDER="host")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-m", "ultra_amd.host_order", "--probe-to", path, "--K", str(K)], env=env,
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise ValueError((r.stderr or r.stdout).strip().splitlines()[-1] if (r.stderr or r.stdout).strip() else "probe failed")
    return load_stages(path, K)


def readout_stages(K=128):
    """(stages, source) of the readout's last product.  ULTRA_READOUT_ORDER selects:
        host (default)   this host's F.linear(x, (1, K)) association: read from the on-disk cache (keyed by CPU model, torch
                         build and BLAS; ULTRA_ORDER_CACHE_DIR, default ~/.cache/ultra_amd) or probed in a helper
                         process (ULTRA_ORDER_PROBE=inprocess: in this one) and cached;
        sequential       one k-ascending fma chain;
        <path>           an order file (save_stages): another host's association, for bit-reproducible scores across hosts.
    Anything that fails falls back to the sequential chain with a warning; the choice is logged once per process."""
    if K in _CACHE:
        return _CACHE[K]
    mode = os.environ.get("ULTRA_READOUT_ORDER", "host")
    stages, source = sequential_stages(K), "sequential"
    try:
        if mode == "host":
            key, text = _host_key(K)
            path = os.path.join(_cache_dir(), "readout_order_%s.json" % key)
            cached = None
            if os.path.exists(path):
                # The key covers CPU model, torch build and BLAS -- not MKL's environment switches (MKL_ENABLE_INSTRUCTIONS,
                # MKL_CBWR, ...), and the directory may be shared by hosts: the cached association is re-checked against
                # this process's F.linear on random rows (cheap: one 4096-row GEMV under the current thread count, where
                # each thread's share may end in a few remainder rows) and probed afresh when it no longer holds.
                try:
                    cached = load_stages(path, K)
                    match = _match_fraction(cached[0], K)
                    if match < 0.9:
                        warnings.warn("ultra_amd: cached readout order %s reproduces only %.1f %% of this host's rows; probing again"
                                      % (order_id(cached[0]), 100 * match))
                        cached = None
                except (OSError, ValueError, KeyError, TypeError):
                    cached = None
            if cached is not None:
                stages, source = cached
                source += " [cached]"
            else:
                try:
                    os.makedirs(_cache_dir(), exist_ok=True)
                except OSError:
                    path = None
                if path and os.environ.get("ULTRA_ORDER_PROBE", "subprocess") == "subprocess":
                    stages, source = _probe_in_subprocess(K, path)
                else:
                    stages, source = probe_host(K)
                    if path:
                        save_stages(path, stages, source, K, text)
        elif mode != "sequential":
            stages, source = load_stages(mode, K)
            source = "file %s (%s)" % (os.path.basename(mode), source)
    except (OSError, ValueError, KeyError, TypeError, IndexError, ArithmeticError, RuntimeError, subprocess.SubprocessError) as exc:
        # (an unreadable cache directory, a tree outside the family, a failed helper process, an in-process probe that trips over
        # torch / numpy: the sequential chain always works -- "anything that fails falls back", as the docstring says; only
        # KeyboardInterrupt / SystemExit / MemoryError pass)
        warnings.warn("ultra_amd: readout summation order '%s' unavailable (%s: %s); using the sequential chain"
                      % (mode, type(exc).__name__, exc))
        stages, source = sequential_stages(K), "sequential (fallback)"
    import logging
    logging.getLogger("ultra_amd").info("readout summation order %s: %s", order_id(stages), source)
    _CACHE[K] = (stages, source)
    return _CACHE[K]


def readout_program(K=128):
    stages, source = readout_stages(K)
    return stages_to_program(stages), source


def describe(K=128):
    """'order-xxxxxxxx: source' of the order in use (bench.py's config.summation_order)."""
    stages, source = readout_stages(K)
    return "%s: %s" % (order_id(stages), source)


def _main():
    import argparse
    ap = argparse.ArgumentParser(description="Probe / save the host BLAS summation order of nn.Linear(K, 1)")
    ap.add_argument("--K", type=int, default=128)
    ap.add_argument("--probe-to", help="probe this host and write the order file (used by readout_stages' helper process)")
    ap.add_argument("--save", help="write the order this host would use (cache or probe) to FILE, for ULTRA_READOUT_ORDER=FILE elsewhere")
    args = ap.parse_args()
    if args.probe_to:
        stages, source = probe_host(args.K)
        save_stages(args.probe_to, stages, source, args.K, _host_key(args.K)[1])
        return
    stages, source = readout_stages(args.K)
    if args.save:
        save_stages(args.save, stages, source, args.K, _host_key(args.K)[1])
    print(describe(args.K))


if __name__ == "__main__":
    _main()
