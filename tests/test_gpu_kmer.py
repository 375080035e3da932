"""GPU parity of the k-mer seed stage (grm::KmerAligner) against the Python restatement (oracle/kmeralign.py,
pinned on the reference's unit test src/c++/test/test_kmeraligner.cpp:149-193)."""
import random

import pytest

from tests import fuzzgen

pytestmark = pytest.mark.gpu

KEYS = ("graph_pos", "score", "cigar")


def gpu_kmer(ctx, graphs, paths, reads, gor, k):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    G.build_kmer_index(paths, k)
    b = ctx.new_batch()
    b.upload(G, reads, gor)
    flags = b.kmer_align()
    res, ops = b.download()
    out = capi.results_to_dicts(res, ops)
    b.close()
    G.close()
    return flags, out


def check(flags, got, want, reads, what):
    n = 0
    for i, (f, g, w) in enumerate(zip(flags, got, want)):
        st = 1 if f & 1 else (2 if f & 4 else 0)
        assert st == w["status"], (what, i, reads[i], f, g, w)
        if st:
            n += 1
            assert all(g[key] == w[key] for key in KEYS) and g["returned_reverse"] == w["used_reverse"], (what, i, reads[i], g, w)
            assert g["mapq"] == w["mapq"] and g["unique"] == w["unique"]
    return n


def test_reference_unit_vectors(gpu_ctx):
    from oracle import kmeralign as ka
    nodes = ["AAAAAAAAAAA", "TTTTTTTT", "GGGGGGGG", "AAAAAAAAAAA"]
    edges = [(0, 1), (0, 2), (0, 3), (1, 3), (2, 3)]
    paths = [[0, 1, 3], [0, 2, 3], [0, 3]]
    reads = ["AAAAAAAATTTTTTTTAAAAAAAA", "TTTTTTAAAAAAAATTTTTTT", "AAAAAGGGGGGGGAAAAAA", "AAAAGGGGGGGGAAAAAA",
             "TTTTTTCCCCCCCCTTTTT", "AAAAAAAAAAAAAAAAAAA"]
    want = [(1, 3, "0[8M]1[8M]3[8M]", 24, False), (1, 4, "0[7M]1[8M]3[6M]", 21, True), (1, 6, "0[5M]2[8M]3[6M]", 19, False),
            (1, 7, "0[4M]2[8M]3[6M]", 18, False), (1, 6, "0[5M]2[8M]3[6M]", 19, True), (2, 0, "0[11M]3[8M]", 19, False)]
    flags, got = gpu_kmer(gpu_ctx, [(nodes, edges)], [paths], reads, None, 10)
    for f, g, (st, pos, cigar, score, rev) in zip(flags, got, want):
        assert (1 if f & 1 else (2 if f & 4 else 0)) == st
        assert (g["graph_pos"], g["cigar"], g["score"], g["returned_reverse"]) == (pos, cigar, score, rev)
    assert ka.port_kmer_align(nodes, paths, reads, 10)[5]["status"] == 2


def test_a_read_beyond_the_stage_limit_is_left_to_the_later_stages(gpu_ctx):
    """One 600-base read in the batch: the stage aligns the others as if it were not there and leaves it without a result or a
    flag (in the cascade it falls through to the graph aligner's general path) -- it used to cost the stage the batch."""
    nodes = ["AAAAAAAAAAA", "TTTTTTTT", "GGGGGGGG", "AAAAAAAAAAA"]
    edges = [(0, 1), (0, 2), (0, 3), (1, 3), (2, 3)]
    paths = [[0, 1, 3], [0, 2, 3], [0, 3]]
    reads = ["AAAAAAAATTTTTTTTAAAAAAAA", "AAAAAGGGGGGGGAAAAAA", "AAAAGGGGGGGGAAAAAA"]
    f0, g0 = gpu_kmer(gpu_ctx, [(nodes, edges)], [paths], reads, None, 10)
    f1, g1 = gpu_kmer(gpu_ctx, [(nodes, edges)], [paths], reads[:1] + ["ACGT" * 150] + reads[1:], None, 10)
    assert list(f1[:1]) + list(f1[2:]) == list(f0) and g1[:1] + g1[2:] == g0
    assert f1[1] == 0 and g1[1]["score"] == 0 and g1[1]["cigar"] == ""


@pytest.mark.parametrize("n_paths", [31, 64, 126])
def test_more_than_30_paths_on_a_graph(gpu_ctx, n_paths):
    """The stage's candidate heap holds paths + 2 entries (KmerAligner.cpp:388-442 replayed in LDS): graphs with up to 126
    paths -- a chain of seven bubbles (128 haplotype paths, 31 / 64 / 126 given) against the restatement."""
    from oracle import kmeralign as ka
    from tests.test_gpu_klib import _bubble_chain
    rng = random.Random(fuzzgen.salted(4100 + n_paths))
    seqs, edges, all_paths = _bubble_chain(rng, 7, 18, 160)
    ps = rng.sample(all_paths, n_paths)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for _ in range(60):
        pseq = "".join(seqs[n] for n in rng.choice(all_paths))
        st = rng.randrange(len(pseq) - 100)
        r = fuzzgen.mutate(rng, pseq[st:st + 100], sub=rng.choice([0.0, 0.01, 0.02]), indel=0.0) or "A"
        if rng.random() < 0.5:
            r = "".join(comp[c] for c in reversed(r))
        reads.append(r)
    want = ka.port_kmer_align(seqs, ps, reads, 16)
    flags, got = gpu_kmer(gpu_ctx, [(seqs, edges)], [ps], reads, None, 16)
    n = check(flags, got, want, reads, "kmer-%d-paths" % n_paths)
    assert n >= 25


def _rand_paths(rng, n_nodes, edges):
    succ = {}
    for f, t in edges:
        succ.setdefault(f, []).append(t)
    roots = [i for i in range(n_nodes) if not any(t == i for _, t in edges)] or [0]
    paths = []
    for _ in range(rng.randint(1, 4)):
        cur = rng.choice(roots)
        p = [cur]
        while cur in succ:
            cur = rng.choice(succ[cur])
            p.append(cur)
        if p not in paths:
            paths.append(p)
    return paths


@pytest.mark.parametrize("k", [10, 16])
def test_kmer_stage_fuzz(gpu_ctx, k):
    from oracle import kmeralign as ka
    from oracle.pathalign import _rc
    rng = random.Random(fuzzgen.salted(77 + k))
    graphs, paths, reads, gor, want = [], [], [], [], []
    for gi in range(150):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=50, max_nodes=6, shape=rng.choice(["del", "bubble", "dag", "longdel"]))
        seqs = [s.replace("X", "N") if rng.random() < 0.5 else s for s in seqs]
        ps = _rand_paths(rng, len(seqs), edges)
        rs = []
        for _ in range(8):
            p = rng.choice(ps)
            pseq = "".join(seqs[n] for n in p)
            L = rng.randint(k, 70)
            st = rng.randrange(max(1, len(pseq) - L + 1))
            r = pseq[st:st + L]
            kind = rng.random()
            if kind < 0.4:
                r = fuzzgen.mutate(rng, r, sub=0.03, indel=0.0)
            elif kind < 0.5:
                r = fuzzgen.mutate(rng, r, sub=0.0, indel=0.03)
            if rng.random() < 0.4:
                r = _rc(r)
            if rng.random() < 0.05:
                r = r.lower()
            rs.append(r or "A")
        graphs.append((seqs, edges))
        paths.append(ps)
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(ka.port_kmer_align(seqs, ps, rs, k))
    flags, got = gpu_kmer(gpu_ctx, graphs, paths, reads, gor, k)
    n = check(flags, got, want, reads, "kmer-fuzz-%d" % k)
    assert n > 300


def test_reference_vector_through_the_cascade_at_k_10(gpu_ctx, checker):
    """The reference's one vector for this aligner (src/c++/test/test_kmeraligner.cpp:149-193, K = 10) through the CASCADE the way
    CompositeAligner runs it (CompositeAligner.cpp:96-118): k-mer stage, filter chain, hand-over on the device, gssw stage on
    what is left.  The five reads the vector maps keep the vector's record (status MAPPED, by the k-mer aligner); the sixth --
    BAD_ALIGN in the vector: two equally good candidates -- goes on and comes out as the reference's gssw aligns it."""
    import numpy as np
    from paragraph_amd import capi
    nodes = ["AAAAAAAAAAA", "TTTTTTTT", "GGGGGGGG", "AAAAAAAAAAA"]
    edges = [(0, 1), (0, 2), (0, 3), (1, 3), (2, 3)]
    paths = [[0, 1, 3], [0, 2, 3], [0, 3]]
    reads = ["AAAAAAAATTTTTTTTAAAAAAAA", "TTTTTTAAAAAAAATTTTTTT", "AAAAAGGGGGGGGAAAAAA", "AAAAGGGGGGGGAAAAAA",
             "TTTTTTCCCCCCCCTTTTT", "AAAAAAAAAAAAAAAAAAA"]
    want = [(3, "0[8M]1[8M]3[8M]", 24, False), (4, "0[7M]1[8M]3[6M]", 21, True), (6, "0[5M]2[8M]3[6M]", 19, False),
            (7, "0[4M]2[8M]3[6M]", 18, False), (6, "0[5M]2[8M]3[6M]", 19, True)]
    G = gpu_ctx.upload_graphs([(nodes, edges)])
    G.set_labels([{(0, 1): ["P"], (1, 3): ["P"], (0, 2): ["Q"], (2, 3): ["Q"], (0, 3): ["R"]}])
    G.build_kmer_index([paths], 10)
    b = gpu_ctx.new_batch()
    b.upload(G, reads, None)
    b.set_fragments(np.arange(len(reads), dtype=np.uint32))
    flags = b.kmer_align()
    # bad_align_frac 0.8 / remove_nonuniq: none of the five full-length unique matches is filtered
    b.count(remove_nonuniq=True, bad_align_frac=0.8)
    b.retire_mapped()
    b.align(capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS)
    b.count(remove_nonuniq=True, bad_align_frac=0.8)
    res, ops, _, sup, _ = b.download_all(want_table=False)
    got = capi.results_to_dicts(res, ops)
    assert [1 if f & 1 else (2 if f & 4 else 0) for f in flags] == [1, 1, 1, 1, 1, 2]
    for g, r, (pos, cigar, score, rev) in zip(got, res, want):
        assert int(r["status"]) & capi.STATUS_KMER_ALIGNER
        assert (g["graph_pos"], g["cigar"], g["score"], g["returned_reverse"]) == (pos, cigar, score, rev)
    assert all(int(s) == 1 for s in sup["status"][:5])
    w = checker.align_batch(nodes, edges, reads[5:])[0]
    assert not (int(res[5]["status"]) & capi.STATUS_KMER_ALIGNER)
    assert all(got[5][k] == w[k] for k in ("graph_pos", "score", "mapq", "unique", "returned_reverse", "cigar")), (got[5], w)
    b.close()
    G.close()
