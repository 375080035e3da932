"""GPU parity of the klib stage (grm::KlibAligner) against the reference's own ksw.c under the restated wrapper
(oracle/_ref) -- or the scalar ksw restatement when _ref is absent; both pinned by tests/test_klib_oracle.py on
src/c++/test/test_klibaligner.cpp:149-193 and src/c++/test/test_align.cpp:38-147."""
import random

import pytest

from tests import fuzzgen
from tests.test_klib_oracle import KA_EXPECT, KA_NODES, KA_PATHS, KA_READS, random_case

pytestmark = pytest.mark.gpu

KEYS = ("graph_pos", "score", "cigar")


def checker():
    from oracle import select
    return select.klib()


def gpu_klib(ctx, graphs, paths, reads, gor, expect_packed=None, active=None):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    G.build_klib_index(paths)
    b = ctx.new_batch()
    b.upload(G, reads, gor)
    if active is not None:
        b.set_active(active)
    flags = b.klib_align()
    res, ops = b.download()
    out = capi.results_to_dicts(res, ops)
    assert G.klib_error() == 0
    if expect_packed is not None:
        assert G.klib_used_packed_kernels() == expect_packed
    b.close()
    G.close()
    return flags, out


def check(flags, got, want, reads, what):
    n = 0
    for i, (f, g, w) in enumerate(zip(flags, got, want)):
        assert not w["ub"]
        st = 1 if f & 1 else (2 if f & 4 else 0)
        assert st == w["status"], (what, i, reads[i], f, g, w)
        if st:
            n += 1
            assert all(g[key] == w[key] for key in KEYS) and g["returned_reverse"] == w["used_reverse"], (what, i, reads[i], g, w)
            assert g["mapq"] == w["mapq"] and g["unique"] == w["unique"]
    return n


def edges_of(paths):
    return sorted({(p[i], p[i + 1]) for p in paths for i in range(len(p) - 1)})


def test_reference_unit_vectors(gpu_ctx):
    flags, got = gpu_klib(gpu_ctx, [(KA_NODES, edges_of(KA_PATHS))], [KA_PATHS], KA_READS, None)
    for f, g, (pos, cigar, score, rev) in zip(flags, got, KA_EXPECT):
        assert f & 1 and g["mapq"] == 60 and g["unique"]
        assert (g["graph_pos"], g["cigar"], g["score"], g["returned_reverse"]) == (pos, cigar, score, rev)


def test_klib_stage_fuzz_bubbles(gpu_ctx):
    chk = checker()
    rng = random.Random(fuzzgen.salted(4242))
    graphs, paths, reads, gor, want = [], [], [], [], []
    for gi in range(120):
        nodes, ps, rs = random_case(rng, 10)
        graphs.append((nodes, edges_of(ps) or [(0, len(nodes) - 1)]))
        paths.append(ps)
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(chk.align(nodes, ps, rs))
    flags, got = gpu_klib(gpu_ctx, graphs, paths, reads, gor)
    n = check(flags, got, want, reads, "klib-bubbles")
    assert n > 800
    assert sum(1 for w in want if w["status"] == 2) > 5


def _rand_paths(rng, n_nodes, edges):
    succ = {}
    for f, t in edges:
        succ.setdefault(f, []).append(t)
    roots = [i for i in range(n_nodes) if not any(t == i for _, t in edges)] or [0]
    paths = []
    for _ in range(rng.randint(1, 5)):
        cur = rng.choice(roots)
        p = [cur]
        while cur in succ:
            cur = rng.choice(succ[cur])
            p.append(cur)
        if p not in paths:
            paths.append(p)
    return paths


def test_klib_stage_fuzz_dags(gpu_ctx):
    from oracle.pathalign import _rc
    chk = checker()
    rng = random.Random(fuzzgen.salted(99))
    graphs, paths, reads, gor, want = [], [], [], [], []
    for gi in range(100):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=60, max_nodes=7, shape=rng.choice(["del", "bubble", "dag", "longdel"]))
        seqs = [s.replace("X", "N") if rng.random() < 0.5 else s for s in seqs]
        ps = _rand_paths(rng, len(seqs), edges)
        rs = []
        for _ in range(8):
            p = rng.choice(ps)
            pseq = "".join(seqs[n] for n in p)
            L = rng.randint(5, 200)
            st = rng.randrange(max(1, len(pseq) - 10))
            r = fuzzgen.mutate(rng, pseq[st:st + L], sub=rng.choice([0.0, 0.03, 0.1]), indel=rng.choice([0.0, 0.02, 0.06])) or "A"
            if rng.random() < 0.4:
                r = _rc(r)
            if rng.random() < 0.05:
                r = r.lower()
            rs.append(r[:250])
        graphs.append((seqs, edges))
        paths.append(ps)
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(chk.align(seqs, ps, rs))
    flags, got = gpu_klib(gpu_ctx, graphs, paths, reads, gor)
    n = check(flags, got, want, reads, "klib-dags")
    assert n > 500


def _bubble_chain(rng, n_bubbles, arm, flank):
    """flank - (a | b) - link - (a | b) - ... - flank: 2^n_bubbles source-to-sink paths"""
    seqs = ["".join(rng.choice("ACGT") for _ in range(flank))]
    edges, layers = [], [[0]]
    for _ in range(n_bubbles):
        a, b = len(seqs), len(seqs) + 1
        seqs += ["".join(rng.choice("ACGT") for _ in range(arm)), "".join(rng.choice("ACGT") for _ in range(arm))]
        link = len(seqs)
        seqs.append("".join(rng.choice("ACGT") for _ in range(12)))
        prev = layers[-1][0]
        edges += [(prev, a), (prev, b), (a, link), (b, link)]
        layers.append([link])
    seqs.append("".join(rng.choice("ACGT") for _ in range(flank)))
    edges.append((layers[-1][0], len(seqs) - 1))
    paths = []
    for m in range(1 << n_bubbles):
        p = [0]
        for k in range(n_bubbles):
            p += [1 + 3 * k + ((m >> k) & 1), 3 + 3 * k]
        paths.append(p + [len(seqs) - 1])
    return seqs, sorted(edges), paths


@pytest.mark.parametrize("n_paths", [31, 64, 126])
def test_more_than_30_paths_on_a_graph(gpu_ctx, n_paths):
    """The stage's candidate heap holds paths + 2 entries (KlibAligner.cpp:388-442): up to 30 paths it lives in registers,
    beyond that the select / pick kernels run with 128-entry heaps -- a chain of seven bubbles (128 haplotype paths, 31 / 64 / 126
    of them given) against the reference's ksw.c under the restated KlibAligner."""
    chk = checker()
    rng = random.Random(fuzzgen.salted(3100 + n_paths))
    seqs, edges, all_paths = _bubble_chain(rng, 7, 18, 160)
    ps = rng.sample(all_paths, n_paths)
    reads = []
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for _ in range(60):
        p = rng.choice(all_paths)  # (also haplotypes that are not among the given paths)
        pseq = "".join(seqs[n] for n in p)
        st = rng.randrange(len(pseq) - 150)
        r = fuzzgen.mutate(rng, pseq[st:st + 150], sub=rng.choice([0.0, 0.01, 0.04]), indel=rng.choice([0.0, 0.0, 0.02]))[:200] or "A"
        if rng.random() < 0.5:
            r = "".join(comp[c] for c in reversed(r))
        reads.append(r)
    want = chk.align(seqs, ps, reads)
    flags, got = gpu_klib(gpu_ctx, [(seqs, edges)], [ps], reads, None)
    n = check(flags, got, want, reads, "klib-%d-paths" % n_paths)
    assert n >= 50


def test_klib_stage_150bp_site(gpu_ctx):
    """Reads of the bench's shape (150 bp, ~500 bp of paths) incl. indel-bearing ones: R = 3 rows per lane."""
    chk = checker()
    rng = random.Random(fuzzgen.salted(7))
    lf = "".join(rng.choice("ACGT") for _ in range(200))
    rf = "".join(rng.choice("ACGT") for _ in range(200))
    alt = "".join(rng.choice("ACGT") for _ in range(60))
    nodes = [lf, alt, alt[:20] + "".join(rng.choice("ACGT") for _ in range(30)), rf]
    ps = [[0, 1, 3], [0, 2, 3], [0, 3]]
    reads = []
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for _ in range(300):
        p = rng.choice(ps)
        seq = "".join(nodes[x] for x in p)
        st = rng.randrange(len(seq) - 150)
        r = fuzzgen.mutate(rng, seq[st:st + 150], sub=rng.choice([0.0, 0.01, 0.05]), indel=rng.choice([0.0, 0.01, 0.03]))[:250] or "A"
        if rng.random() < 0.5:
            r = "".join(comp[c] for c in reversed(r))
        reads.append(r)
    want = chk.align(nodes, ps, reads)
    flags, got = gpu_klib(gpu_ctx, [(nodes, edges_of(ps))], [ps], reads, None)
    n = check(flags, got, want, reads, "klib-150")
    assert n >= 290


@pytest.mark.parametrize("general", [False, True])
def test_a_read_beyond_the_stage_limit_is_left_to_the_later_stages(gpu_ctx, monkeypatch, general):
    """One 700-base read in the batch (packed and general kernels): the stage aligns the others as if it were not there and
    leaves it without a result or a flag -- in the cascade it falls through to the graph aligner's general path.  It used to
    cost the stage the whole batch (and the workflow that site)."""
    if general:
        monkeypatch.setenv("PG_KLIB_GENERAL", "1")
    rng = random.Random(fuzzgen.salted(77))
    lf = "".join(rng.choice("ACGT") for _ in range(400))
    rf = "".join(rng.choice("ACGT") for _ in range(400))
    alt = "".join(rng.choice("ACGT") for _ in range(60))
    nodes = [lf, alt, rf]
    ps = [[0, 1, 2], [0, 2]]
    reads = []
    for _ in range(40):
        seq = "".join(nodes[x] for x in rng.choice(ps))
        st = rng.randrange(len(seq) - 150)
        reads.append(fuzzgen.mutate(rng, seq[st:st + 150], sub=0.01, indel=0.005)[:200] or "A")
    long_read = (lf + alt + rf)[:700]
    f0, g0 = gpu_klib(gpu_ctx, [(nodes, edges_of(ps))], [ps], reads, None)
    mixed = reads[:7] + [long_read] + reads[7:]
    f1, g1 = gpu_klib(gpu_ctx, [(nodes, edges_of(ps))], [ps], mixed, None)
    assert list(f1[:7]) + list(f1[8:]) == list(f0) and g1[:7] + g1[8:] == g0
    assert f1[7] == 0 and g1[7]["score"] == 0 and g1[7]["cigar"] == ""
    assert sum(1 for f in f0 if f & 1) >= 35


def test_klib_stage_long_reads(gpu_ctx):
    """300..512 bp reads (the general kernels where a path is shorter than a read: R = 5..8 rows per lane, 8 direction bytes
    per lane per step; the packed ones with C = 20..32 rows per lane otherwise)."""
    chk = checker()
    rng = random.Random(fuzzgen.salted(11))
    graphs, paths, reads, gor, want = [], [], [], [], []
    for gi in range(12):
        seqs, edges, rs = fuzzgen.long_read_case(rng, 6)
        last = len(seqs) - 1
        ps = [[0, i, last] for i in range(1, last)] + ([[0, last]] if (0, last) in edges else [])
        rs = [r for r in rs] + [fuzzgen.mutate(rng, (seqs[0] + seqs[1] + seqs[last])[:rng.randint(300, 512)], sub=0.02, indel=0.01)[:512]]
        graphs.append((seqs, edges))
        paths.append(ps)
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(chk.align(seqs, ps, rs))
    flags, got = gpu_klib(gpu_ctx, graphs, paths, reads, gor)
    n = check(flags, got, want, reads, "klib-long")
    assert n > 60 and max(len(r) for r in reads) > 450


def test_klib_after_kmer_keeps_results(gpu_ctx):
    """Cascade use: k-mer stage first, klib only on what it left unmapped, earlier results kept."""
    import numpy as np
    from paragraph_amd import capi
    chk = checker()
    from oracle import kmeralign as ka
    reads = KA_READS + ["AAAAAAAATTTTCTTTAAAAAAAA", "AAAAAGGGGGAAAAAA"]
    G = gpu_ctx.upload_graphs([(KA_NODES, edges_of(KA_PATHS))])
    G.build_kmer_index([KA_PATHS], 10)
    G.build_klib_index([KA_PATHS])
    b = gpu_ctx.new_batch()
    b.upload(G, reads, None)
    f1 = b.kmer_align()
    active = np.array([0 if f & 1 else 1 for f in f1], dtype=np.uint8)
    b.set_active(active)
    f2 = b.klib_align(capi.AF_KEEP_RESULTS)
    res, ops = b.download()
    out = capi.results_to_dicts(res, ops)
    wk = ka.port_kmer_align(KA_NODES, KA_PATHS, reads, 10)
    wl = chk.align(KA_NODES, KA_PATHS, reads)
    assert active.sum() >= 1
    for i in range(len(reads)):
        w = wl[i] if active[i] else wk[i]
        if active[i]:
            assert (1 if f2[i] & 1 else (2 if f2[i] & 4 else 0)) == w["status"]
        if w["status"]:
            assert all(out[i][k] == w[k] for k in KEYS), (i, out[i], w)
    b.close()
    G.close()


def test_klib_after_kmer_hand_over_on_the_device(gpu_ctx):
    """The same cascade with the hand-over decided on the device (pg_batch_retire_mapped after the k-mer stage's count pass): the
    klib stage's work items are re-made from the per-graph counts of the reads still active; records equal the host-mask run."""
    import numpy as np
    from paragraph_amd import capi
    reads = (KA_READS + ["AAAAAAAATTTTCTTTAAAAAAAA", "AAAAAGGGGGAAAAAA"]) * 40
    edges = edges_of(KA_PATHS)
    out = {}
    for on_device in (False, True):
        G = gpu_ctx.upload_graphs([(KA_NODES, edges)])
        G.set_labels([{e: ["L%d" % (k % 3)] for k, e in enumerate(edges)}])
        G.build_kmer_index([KA_PATHS], 10)
        G.build_klib_index([KA_PATHS])
        b = gpu_ctx.new_batch()
        b.upload(G, reads, None)
        b.set_fragments(np.arange(len(reads), dtype=np.uint32))
        f1 = b.kmer_align()
        b.count(remove_nonuniq=True, bad_align_frac=0.8)
        active = np.ones(len(reads), dtype=bool)
        if on_device:
            b.retire_mapped()
        else:
            _, sup, _ = b.download_counts(want_table=False)
            active &= ~(((f1 & 1) != 0) & (sup["status"] == 1))
            b.set_active(active)
        f2 = b.klib_align(capi.AF_KEEP_RESULTS)
        b.count(remove_nonuniq=True, bad_align_frac=0.8)
        if on_device:
            b.retire_mapped()
        else:
            _, sup, _ = b.download_counts(want_table=False)
            active &= ~(((f2 & 1) != 0) & (sup["status"] == 1))
            b.set_active(active)
        b.align(capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS)
        b.count(remove_nonuniq=True, bad_align_frac=0.8)
        res, ops, table, sup, _ = b.download_all()
        out[on_device] = (f1.copy(), f2.copy(), capi.results_to_dicts(res, ops), table, sup, active.copy())
        b.close()
        G.close()
    (a1, a2, ra, ta, sa, act), (b1, b2, rb, tb, sb, _) = out[False], out[True]
    assert np.array_equal(a1, b1) and np.array_equal(ta, tb) and np.array_equal(sa["status"], sb["status"])
    ran = (a1 & 1) == 0  # the klib stage's flags agree wherever it ran in both (a read the k-mer stage retired is not touched by it)
    assert ran.sum() >= 40 and np.array_equal(a2[ran], b2[ran])
    assert 0 < act.sum() < len(reads)  # something reached the gssw stage, something did not
    for i in range(len(reads)):
        assert all(ra[i][k] == rb[i][k] for k in KEYS + ("status",)), (i, ra[i], rb[i])


# ---- the packed two-strand kernels (reads <= 250 bases, every path at least as long as the longest read) -------------------

def _site(rng, n_alt, flank):
    """LF, n_alt ALT nodes (some equal to each other or to a piece of a flank: ties between paths), RF; paths LF-ALT-RF and
    LF-RF.  Flanks of `flank` bases keep every path longer than the reads."""
    lf = fuzzgen.rand_seq(rng, flank + rng.randint(0, 40), rng.choice(["rand", "rand", "rand", "period", "two"]))
    rf = fuzzgen.rand_seq(rng, flank + rng.randint(0, 40), rng.choice(["rand", "rand", "rand", "period", "two"]))
    alts = []
    for _ in range(n_alt):
        u = rng.random()
        if alts and u < 0.25:
            alts.append(rng.choice(alts))  # a second node with the same sequence: equal-score candidates on different paths
        elif u < 0.4:
            alts.append(fuzzgen.mutate(rng, rng.choice(alts), sub=0.05, indel=0.02) or "A" if alts else fuzzgen.rand_seq(rng, 30))
        else:
            alts.append(fuzzgen.rand_seq(rng, rng.randint(1, 120)))
    nodes = [lf] + alts + [rf]
    if rng.random() < 0.3:
        i = rng.randrange(len(nodes))
        s = list(nodes[i])
        for _ in range(rng.randint(1, 3)):
            s[rng.randrange(len(s))] = "N"
        nodes[i] = "".join(s)
    last = len(nodes) - 1
    paths = [[0, i, last] for i in range(1, last)] + [[0, last]]
    return nodes, paths


def _site_reads(rng, nodes, paths, n, max_len):
    from oracle.pathalign import _rc
    reads = []
    for _ in range(n):
        u = rng.random()
        if u < 0.04:
            r = fuzzgen.rand_seq(rng, rng.randint(1, max_len))  # unrelated
        else:
            seq = "".join(nodes[x] for x in rng.choice(paths))
            L = rng.choice([rng.randint(1, 40), rng.randint(30, max_len), 150, 150, max_len])
            st = rng.randrange(max(1, len(seq) - L + 1))
            r = seq[st:st + L]
            r = fuzzgen.mutate(rng, r, sub=rng.choice([0.0, 0.0, 0.01, 0.05, 0.15]), indel=rng.choice([0.0, 0.0, 0.01, 0.04]),
                               nrate=rng.choice([0.0, 0.0, 0.02])) or "A"
            if rng.random() < 0.15 and len(r) > 60:  # a long deletion / insertion inside the read
                cut = rng.randrange(20, len(r) - 20)
                r = r[:cut] + (fuzzgen.rand_seq(rng, rng.randint(1, 30)) if rng.random() < 0.5 else "") + r[cut + rng.randint(0, 40):]
            if rng.random() < 0.1:  # clipped ends
                r = fuzzgen.rand_seq(rng, rng.randint(1, 25)) + r + fuzzgen.rand_seq(rng, rng.randint(0, 25))
        r = r[:max_len] or "A"
        if rng.random() < 0.5:
            r = _rc(r)
        if rng.random() < 0.05:
            r = r.lower()
        reads.append(r)
    return reads


def _packed_case(seed, n_graphs, reads_per_graph, max_len, flank):
    chk = checker()
    rng = random.Random(fuzzgen.salted(seed))
    graphs, paths, reads, gor, want = [], [], [], [], []
    for gi in range(n_graphs):
        nodes, ps = _site(rng, rng.randint(1, 5), flank)
        rs = _site_reads(rng, nodes, ps, reads_per_graph, max_len)
        graphs.append((nodes, edges_of(ps)))
        paths.append(ps)
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(chk.align(nodes, ps, rs))
    return graphs, paths, reads, gor, want


@pytest.mark.parametrize("seed,max_len,flank", [(501, 150, 160), (502, 250, 260), (503, 100, 130), (504, 33, 40), (505, 380, 400),
                                                (506, 512, 520)])
def test_klib_packed_kernels_fuzz(gpu_ctx, seed, max_len, flank):
    """Mixed read lengths (several rows-per-lane classes in one batch), ties between paths, N, lower case, long gaps, clips."""
    graphs, paths, reads, gor, want = _packed_case(seed, 40 if max_len <= 250 else 12, 60 if max_len <= 250 else 40, max_len, flank)
    flags, got = gpu_klib(gpu_ctx, graphs, paths, reads, gor, expect_packed=True)
    n = check(flags, got, want, reads, "klib-packed-%d" % seed)
    assert n > 0.85 * len(reads)
    assert sum(1 for w in want if w["status"] == 2) > 3


@pytest.mark.parametrize("seed,max_len,flank", [(611, 200, 210), (612, 330, 340)])
def test_klib_packed_and_general_kernels_agree(gpu_ctx, monkeypatch, seed, max_len, flank):
    """The same batch through both kernel sets (PG_KLIB_GENERAL forces the general one): identical results."""
    graphs, paths, reads, gor, want = _packed_case(seed, 25 if max_len <= 250 else 10, 40, max_len, flank)
    f1, g1 = gpu_klib(gpu_ctx, graphs, paths, reads, gor, expect_packed=True)
    monkeypatch.setenv("PG_KLIB_GENERAL", "1")
    f2, g2 = gpu_klib(gpu_ctx, graphs, paths, reads, gor, expect_packed=False)
    assert list(f1) == list(f2)
    for a, b, f in zip(g1, g2, f1):
        if f & 5:
            assert all(a[k] == b[k] for k in KEYS + ("mapq", "unique", "returned_reverse")), (a, b)
    check(f1, g1, want, reads, "klib-both")


def test_klib_packed_kernels_active_subset(gpu_ctx):
    """Cascade use at scale: only the reads an earlier stage left are planned into wavefronts; the others keep their flags."""
    import numpy as np
    graphs, paths, reads, gor, want = _packed_case(707, 30, 50, 150, 160)
    rng = random.Random(5)
    active = np.array([1 if rng.random() < 0.4 else 0 for _ in reads], dtype=np.uint8)
    flags, got = gpu_klib(gpu_ctx, graphs, paths, reads, gor, expect_packed=True, active=active)
    idx = [i for i in range(len(reads)) if active[i]]
    check([flags[i] for i in idx], [got[i] for i in idx], [want[i] for i in idx], [reads[i] for i in idx], "klib-active")


def test_klib_config2_8192_reads(gpu_ctx):
    """The stage probe's workload (BASELINE configs[1] reads, the two paths of the DEL graph) against the reference's ksw.c:
    every read's status, position, CIGAR, score, strand and MAPQ."""
    from paragraph_amd import synth
    chk = checker()
    n = 8192
    site, arr = synth.config2_reads_packed(n, read_len=150, seed=7)
    raw = arr.tobytes()
    reads = [raw[i * 150:(i + 1) * 150].decode() for i in range(n)]
    ps = [[0, 1, 2], [0, 2]]
    want = []
    for i in range(0, n, 2048):
        want.extend(chk.align(site.seqs, ps, reads[i:i + 2048]))
    flags, got = gpu_klib(gpu_ctx, [(site.seqs, site.edges)], [ps], reads, None, expect_packed=True)
    mapped = check(flags, got, want, reads, "klib-config2")
    assert mapped > 0.99 * n


def test_klib_cigar_pool_is_used_and_its_exhaustion_is_loud(gpu_ctx, monkeypatch):
    """Packed finish kernel: a candidate's CIGAR starts in a 24-entry slot of its own and moves to a full-size slot of the pool
    when it outgrows it (reads with a dozen indels).  Results equal the checker's; with the pool taken away (PG_KLIB_CIG_POOL=0)
    the stage says so through pg_graphs_klib_error (bit 1) instead of returning a cut CIGAR."""
    from oracle.pathalign import _rc
    chk = checker()
    rng = random.Random(fuzzgen.salted(811))
    nodes, ps = _site(rng, 3, 330)
    nodes = [n.replace("N", "A") for n in nodes]
    reads = []
    for _ in range(200):
        seq = "".join(nodes[x] for x in rng.choice(ps))
        st = rng.randrange(max(1, len(seq) - 300))
        r = list(seq[st:st + 300])
        for cut in sorted(rng.sample(range(15, len(r) - 15, 16), 14), reverse=True):  # 14 one-base indels, 16+ bases apart
            if rng.random() < 0.5:
                del r[cut]
            else:
                r.insert(cut, rng.choice("ACGT"))
        r = "".join(r)[:320]
        reads.append(_rc(r) if rng.random() < 0.5 else r)
    want = chk.align(nodes, ps, reads)
    # (that the data reaches the pool at all; how many reads keep a dozen indels in their best alignment depends on the salt: 18 - 45 of 120)
    assert sum(1 for w in want if w["status"] == 1 and w["cigar"].count("I") + w["cigar"].count("D") >= 12) > 12
    flags, got = gpu_klib(gpu_ctx, [(nodes, edges_of(ps))], [ps], reads, None, expect_packed=True)
    check(flags, got, want, reads, "klib-pool")
    monkeypatch.setenv("PG_KLIB_CIG_POOL", "0")
    G = gpu_ctx.upload_graphs([(nodes, edges_of(ps))])
    G.build_klib_index([ps])
    b = gpu_ctx.new_batch()
    b.upload(G, reads, None)
    b.klib_align()
    assert G.klib_error() & 2
    b.close()
    G.close()
