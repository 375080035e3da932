"""Randomized graphs/reads for differential tests (adversarial: low-complexity sequences, near-identical
branches, short nodes, N bases, indels)."""
import random


def rand_seq(rng, n, mode=None):
    mode = mode if mode is not None else rng.choice(["rand", "rand", "period", "homo", "two"])
    if mode == "rand":
        return "".join(rng.choice("ACGT") for _ in range(n))
    if mode == "homo":
        return rng.choice("ACGT") * n
    if mode == "two":
        ab = rng.sample("ACGT", 2)
        return "".join(rng.choice(ab) for _ in range(n))
    p = rng.randint(1, 4)
    unit = "".join(rng.choice("ACGT") for _ in range(p))
    return (unit * (n // p + 1))[:n]


def mutate(rng, s, sub=0.03, indel=0.01, nrate=0.0):
    out = []
    for c in s:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            out.append(rng.choice("ACGT"))
        if rng.random() < sub:
            c = rng.choice("ACGT")
        if nrate and rng.random() < nrate:
            c = "N"
        out.append(c)
    return "".join(out)


def rand_graph(rng, max_nodes=7, max_len=40, shape=None):
    """Returns (node_seqs, edges). Node ids are topologically ordered."""
    shape = shape or rng.choice(["del", "ins", "bubble", "dag", "dag", "chain", "longdel"])
    if shape in ("del", "ins"):
        lens = [rng.randint(1, max_len) for _ in range(3)]
        seqs = [rand_seq(rng, n) for n in lens]
        return seqs, [(0, 1), (0, 2), (1, 2)]
    if shape == "bubble":
        a = rand_seq(rng, rng.randint(1, max_len))
        b = mutate(rng, a, sub=0.1, indel=0.05) or "A"
        return [rand_seq(rng, rng.randint(1, max_len)), a, b, rand_seq(rng, rng.randint(1, max_len))], \
            [(0, 1), (0, 2), (1, 3), (2, 3)] + ([(0, 3)] if rng.random() < 0.5 else [])
    if shape == "chain":
        n = rng.randint(1, max_nodes)
        return [rand_seq(rng, rng.randint(1, max_len)) for _ in range(n)], [(i, i + 1) for i in range(n - 1)]
    if shape == "longdel":
        seqs = ["X"] + [rand_seq(rng, rng.randint(1, max_len)) for _ in range(4)] + ["X"]
        return seqs, [(0, 1), (0, 3), (1, 4), (1, 2), (3, 4), (3, 5), (4, 5)]
    n = rng.randint(2, max_nodes)
    mode = rng.choice([None, None, "homo", "period"])
    seqs = [rand_seq(rng, rng.randint(1, max_len), mode) for _ in range(n)]
    edges = set()
    for t in range(1, n):
        k = rng.randint(0 if rng.random() < 0.15 else 1, min(t, 3))
        for f in rng.sample(range(t), k):
            edges.add((f, t))
    return seqs, sorted(edges)


def rand_path_seq(rng, seqs, edges):
    succ = {}
    for f, t in edges:
        succ.setdefault(f, []).append(t)
    cur = rng.randrange(len(seqs))
    out = [seqs[cur]]
    while cur in succ and rng.random() < 0.9:
        cur = rng.choice(succ[cur])
        out.append(seqs[cur])
    return "".join(out)


def rand_read(rng, seqs, edges, min_len=8, max_len=120):
    kind = rng.random()
    L = rng.randint(min_len, max_len)
    if kind < 0.08:
        r = rand_seq(rng, L)
    else:
        p = rand_path_seq(rng, seqs, edges).replace("X", "")
        if len(p) < 2:
            p = rand_seq(rng, L)
        st = rng.randrange(max(1, len(p) - 1))
        r = p[st:st + L]
        r = mutate(rng, r, sub=rng.choice([0.0, 0.02, 0.1]), indel=rng.choice([0.0, 0.0, 0.02, 0.08]),
                   nrate=rng.choice([0.0, 0.0, 0.0, 0.05]))
        if rng.random() < 0.3:  # soft-clipped ends
            r = rand_seq(rng, rng.randint(0, 10)) + r + rand_seq(rng, rng.randint(0, 10))
    if not r:
        r = "A"
    if rng.random() < 0.5:
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        r = "".join(comp.get(c, "N") for c in reversed(r))
    if rng.random() < 0.02:
        r = r.lower()
    return r


def cases(seed, n_graphs, reads_per_graph, **kw):
    rng = random.Random(seed)
    for _ in range(n_graphs):
        seqs, edges = rand_graph(rng, **kw)
        reads = [rand_read(rng, seqs, edges) for _ in range(reads_per_graph)]
        yield seqs, edges, reads


def rand_labels(rng, edges, max_labels=4):
    """Random edge labels: {(from,to): [names]} + ordered label list (names chosen so that string order differs
    from creation order)."""
    names = rng.sample(["REF", "ALT", "DEL", "INS", "P", "Q", "D", "Z1", "A0"], rng.randint(1, max_labels))
    out = {}
    for e in edges:
        ls = [n for n in names if rng.random() < 0.5]
        if ls:
            out[tuple(e)] = ls
    return out, sorted(names)


def rand_path_labels(rng, n_nodes, edges, n_labels):
    """Many labels the way haplotypes make them: every label names one random source-to-sink walk of the graph and sits on all
    of its edges.  -> {(from,to): [names]}, sorted name list (names L000, L001, ...)."""
    succ = {}
    for f, t in edges:
        succ.setdefault(f, []).append(t)
    starts = [n for n in range(n_nodes) if n in succ and not any(t == n for _, t in edges)] or [min(succ)] if succ else []
    names = ["L%03d" % i for i in range(n_labels)]
    out = {}
    for name in names:
        if not starts:
            break
        n = rng.choice(starts)
        while n in succ:
            t = rng.choice(succ[n])
            out.setdefault((n, t), []).append(name)
            n = t
    return out, names


def rand_fragments(rng, n):
    """Fragment ids: mostly pairs, some singletons, a few fragments with 3 reads."""
    ids = []
    fid = 0
    while len(ids) < n:
        k = rng.choice([1, 2, 2, 2, 3])
        ids.extend([fid] * k)
        fid += 1
    ids = ids[:n]
    rng.shuffle(ids)
    return ids


def long_read_case(rng, n_reads=6):
    """Graphs whose paths are long enough for 251..512 bp reads that overflow gssw's byte mode (score >= 251), incl.
    scores right at the 251..255 boundary and repeated flanks (ties between nodes)."""
    n_mid = rng.randint(1, 3)
    lf = rand_seq(rng, rng.randint(150, 300))
    rf = lf if rng.random() < 0.25 else rand_seq(rng, rng.randint(150, 300))
    mids = []
    for _ in range(n_mid):
        mids.append(mids[0] if mids and rng.random() < 0.3 else rand_seq(rng, rng.randint(1, 120)))
    seqs = [lf] + mids + [rf]
    last = len(seqs) - 1
    edges = [(0, i + 1) for i in range(n_mid)] + [(i + 1, last) for i in range(n_mid)]
    if rng.random() < 0.5:
        edges.append((0, last))
    edges = sorted(set(edges))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for _ in range(n_reads):
        mid = rng.choice(range(n_mid + 1))
        path = lf + (mids[mid] if mid < n_mid else "") + rf if (mid < n_mid or (0, last) in edges) else lf + mids[0] + rf
        L = rng.choice([rng.randint(251, 262), rng.randint(251, 512)])
        st = rng.randrange(max(1, len(path) - L + 1))
        r = path[st:st + L]
        r = mutate(rng, r, sub=rng.choice([0.0, 0.0, 0.005, 0.03]), indel=rng.choice([0.0, 0.0, 0.004]))[:512] or "A"
        if rng.random() < 0.5:
            r = "".join(comp.get(c, "N") for c in reversed(r))
        reads.append(r)
    return seqs, edges, reads


def salted(seed):
    """Seeds of the GPU fuzz tests: PG_SEED_SALT=<n> shifts all of them, so `for n in ...: PG_SEED_SALT=$n pytest -m gpu -k fuzz`
    runs the same comparisons on fresh random inputs (tools/stress_gpu.sh)."""
    import os
    return seed + 7919 * int(os.environ.get("PG_SEED_SALT", "0"))


def multi_equal(got, want):
    """the four alignsEndAtMultNodes flags of a pg_result against the checker's: all four, or -- a record of the lean gssw stage whose
    `other_fwd_skipped` is set -- all but the forward-graph fill of the strand that was not returned (it did not run; its bit reads 0)"""
    gm, wm = list(got["multi"]), list(want["multi"])
    if got.get("other_fwd_skipped"):
        k = 0 if got["returned_reverse"] else 1
        if gm[k] != 0:
            return False
        gm[k] = wm[k]
    return gm == wm


def multi_equal_known(a, b):
    """two pg_results of the SAME read from two runs of the lean gssw stage: the multi flags of the fills BOTH ran (the stage runs the
    forward-graph fill of the strand it does not return only where the record needs it -- and for a run's last pair without a partner,
    which depends on the batch's other reads)"""
    am, bm = list(a["multi"]), list(b["multi"])
    if a.get("other_fwd_skipped") or b.get("other_fwd_skipped"):
        k = 0 if a["returned_reverse"] else 1
        am[k] = bm[k] = 0
    return am == bm
