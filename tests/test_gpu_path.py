"""GPU parity of the exact path matching stage (grm::PathAligner) and of the two-stage cascade
path -> (filter) -> gssw (CompositeAligner with --path-sequence-matching)."""
import random

import numpy as np
import pytest

from tests import fuzzgen

pytestmark = pytest.mark.gpu

PKEYS = ("graph_pos", "score", "mapq", "unique", "cigar")


def path_checker():
    from oracle import select
    return select.path_align()


def gpu_path(ctx, graphs, reads, gor, k):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    G.build_path_index(k)
    b = ctx.new_batch()
    b.upload(G, reads, gor)
    flags = b.path_align()
    res, ops = b.download()
    out = capi.results_to_dicts(res, ops)
    b.close()
    G.close()
    return flags, out


def test_reference_unit_vectors(gpu_ctx):
    """src/c++/test/test_pathaligner.cpp:37-144 (k = 16)."""
    g1 = (["AAAAAAAAA", "CCCC", "GGGGGGGGG"], [(0, 1), (0, 2), (1, 2)])
    reads = ["AAAAAAAAGGGGGGGG", "CCCCCCCCTTTTTTTT", "AAAAAAAACCCCGGGG", "CCCCGGGGTTTTTTTT", "AAAAAAAAGGGGGGGGG",
             "CCCCCCCCCTTTTTTTTT"]
    want = [(1, "0[8M]2[8M]", 16, False), (1, "0[8M]2[8M]", 16, True), (1, "0[8M]1[4M]2[4M]", 16, False),
            (1, "0[8M]1[4M]2[4M]", 16, True), (1, "0[8M]2[9M]", 17, False), (0, "0[9M]2[9M]", 18, True)]
    flags, got = gpu_path(gpu_ctx, [g1], reads, None, 16)
    for f, g, (pos, cigar, score, rev) in zip(flags, got, want):
        assert f & 1 and g["by_path_aligner"]
        assert (g["graph_pos"], g["cigar"], g["score"], g["returned_reverse"], g["mapq"]) == (pos, cigar, score, rev, 60)
    g2 = (["GGGGGGGGGGGG", "CCCCCCCCCCCCCCCC", "GGGGGGGGGGGGGTGGG"], [(0, 1), (0, 2), (1, 2)])
    flags, got = gpu_path(gpu_ctx, [g2], ["CCCCCCCCCCCCGGGGGGGGGGGG"], None, 16)
    assert flags[0] & 1 and got[0]["graph_pos"] == 4 and got[0]["cigar"] == "1[12M]2[12M]" and got[0]["score"] == 24
    assert got[0]["mapq"] == 0 and not got[0]["unique"] and not got[0]["returned_reverse"]


def _path_reads(rng, seqs, edges, k, n):
    from oracle import pathalign as pa
    reads = []
    for _ in range(n):
        p = fuzzgen.rand_path_seq(rng, seqs, edges)
        L = rng.randint(k, 90)
        st = rng.randrange(max(1, len(p)))
        r = p[st:st + L]
        if rng.random() < 0.3:
            r = fuzzgen.mutate(rng, r, sub=0.02, indel=0.0)
        if rng.random() < 0.4:
            r = pa._rc(r)
        if rng.random() < 0.1:
            r = fuzzgen.rand_seq(rng, 3) + r
        reads.append(r or "A")
    return reads


@pytest.mark.parametrize("k,index_on_device", [(8, False), (16, False), (32, False), (16, True), (32, True)])
def test_path_stage_fuzz(gpu_ctx, k, index_on_device, monkeypatch):
    """index_on_device: the k-mer table, node pool and presence filter made by pg_index_build_kernel (the default; queued by the
    set's first path stage on its seed stream) or by the host enumerator (PG_PATH_INDEX_HOST=1) -- the same records either way."""
    if index_on_device:
        monkeypatch.delenv("PG_PATH_INDEX_HOST", raising=False)
    else:
        monkeypatch.setenv("PG_PATH_INDEX_HOST", "1")
    check = path_checker()
    rng = random.Random(fuzzgen.salted(1000 + k))
    graphs, reads, gor, want = [], [], [], []
    for gi in range(250):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=60, max_nodes=6)
        rs = _path_reads(rng, seqs, edges, k, 8)
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(check(seqs, edges, rs, k))
    flags, got = gpu_path(gpu_ctx, graphs, reads, gor, k)
    n_mapped = 0
    for i, (f, g, w) in enumerate(zip(flags, got, want)):
        assert bool(f & 1) == bool(w["status"]), (i, reads[i], graphs[gor[i]], g, w)
        assert bool(f & 2) == w["anchored"], (i, reads[i], g, w)
        if w["status"]:
            n_mapped += 1
            assert all(g[key] == w[key] for key in PKEYS) and g["returned_reverse"] == w["is_graph_reverse"], \
                (i, reads[i], graphs[gor[i]], g, w)
    assert n_mapped > 200


def test_cascade_path_then_gssw(gpu_ctx, checker):
    """CompositeAligner(path=true, graph=true) with the NonUniq + BadAlign filter chain
    (CompositeAligner.cpp:78-176): path-mapped reads that pass the filter keep the path alignment, everything
    else is re-aligned by the gssw stage."""
    from paragraph_amd import capi, synth
    check = path_checker()
    site, reads = synth.config2_reads(2048, read_len=150, seed=11)
    k = 32
    G = gpu_ctx.upload_graphs([(site.seqs, site.edges)])
    G.build_path_index(k)
    b = gpu_ctx.new_batch()
    b.upload(G, reads)
    flags = b.path_align()
    res, _ = b.download()
    # filter after the path stage: NonUniq (remove_nonuniq) ; BadAlign can never fire on a full-length match
    keep_path = (flags & 1).astype(bool) & (res["is_unique"] != 0)
    b.set_active(~keep_path)
    b.align(capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS)
    res, ops = b.download()
    got = capi.results_to_dicts(res, ops)
    wp = check(site.seqs, site.edges, reads, k)
    wg = checker.align_batch(site.seqs, site.edges, reads, threads=8)
    n_path = 0
    for i, g in enumerate(got):
        if wp[i]["status"] and wp[i]["unique"]:
            n_path += 1
            assert g["by_path_aligner"] and all(g[key] == wp[i][key] for key in PKEYS), (i, g, wp[i])
            assert g["returned_reverse"] == wp[i]["is_graph_reverse"]
        else:
            assert not g["by_path_aligner"]
            assert all(g[key] == wg[i][key] for key in PKEYS + ("returned_reverse",)), (i, g, wg[i])
    assert 200 < n_path < 1200
    b.close()
    G.close()


def _cascade(ctx, graphs, labels, reads, gor, k, on_device):
    """path stage -> count (filter chain) -> hand-over -> gssw stage on what is left -> count; the hand-over either as a host
    loop over downloaded flags + supports (pg_batch_set_active) or on the device (pg_batch_retire_mapped)."""
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    G.set_labels(labels)
    G.build_path_index(k)
    b = ctx.new_batch()
    b.upload(G, reads, gor)
    b.set_fragments(np.arange(len(reads), dtype=np.uint32))
    flags = b.path_align()
    b.count(remove_nonuniq=True, bad_align_frac=0.8)
    if on_device:
        b.retire_mapped()
    else:
        _, sup, _ = b.download_counts(want_table=False)
        b.set_active(~(((flags & 1) != 0) & (sup["status"] == 1)))
    b.align(capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS)
    b.count(remove_nonuniq=True, bad_align_frac=0.8)
    res, ops, table, sup, path = b.download_all()
    res2, ops2 = b.download()
    table2, sup2, path2 = b.download_counts()
    # the one-wait download hands back what the separate calls do
    assert np.array_equal(res, res2) and np.array_equal(ops, ops2) and np.array_equal(table, table2)
    assert np.array_equal(sup, sup2) and np.array_equal(path, path2)
    out = capi.results_to_dicts(res, ops)
    b.close()
    G.close()
    return flags, out, sup, table


def test_cascade_hand_over_on_the_device(checker):
    """CompositeAligner's hand-over (CompositeAligner.cpp:78-176) decided on the device: the records, the count-path outcome of
    every read and the site tables equal those of the host-mediated hand-over -- on one hot graph (most reads retire at the path
    stage: the gssw stage's work items are re-made from the per-graph counts) and on 300 small graphs, with a workspace small
    enough that the re-made plan spans many chunks -- and the gssw-stage reads equal the reference's alignments."""
    from paragraph_amd import capi, synth
    check = path_checker()
    site, reads = synth.config2_reads(6000, read_len=150, seed=31)
    graphs, labels, gor = [(site.seqs, site.edges)], [site.labels], [0] * len(reads)
    reads = list(reads)
    rng = random.Random(fuzzgen.salted(77))
    for gi in range(300):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=60, max_nodes=6)
        rs = _path_reads(rng, seqs, edges, 32, 8)
        graphs.append((seqs, edges))
        labels.append(fuzzgen.rand_labels(rng, edges)[0])
        reads.extend(rs)
        gor.extend([gi + 1] * len(rs))
    out = {}
    for on_device in (False, True):
        ctx = capi.Context(0, workspace_bytes=64 << 20)
        out[on_device] = _cascade(ctx, graphs, labels, reads, gor, 32, on_device)
        ctx.close()
    (f0, r0, s0, t0), (f1, r1, s1, t1) = out[False], out[True]
    assert np.array_equal(f0, f1) and np.array_equal(t0, t1)
    assert np.array_equal(s0["status"], s1["status"]) and np.array_equal(s0["label_mask"], s1["label_mask"]) and np.array_equal(s0["n_path"], s1["n_path"])
    for a, b in zip(r0, r1):
        assert all(a[key] == b[key] for key in PKEYS + ("returned_reverse", "by_path_aligner", "status")), (a, b)
    wp = check(site.seqs, site.edges, reads[:6000], 32)
    wg = checker.align_batch(site.seqs, site.edges, reads[:6000], threads=8)
    n_path = 0
    for i, g in enumerate(r1[:6000]):
        if wp[i]["status"] and wp[i]["unique"]:
            n_path += 1
            assert g["by_path_aligner"] and all(g[key] == wp[i][key] for key in PKEYS), (i, g, wp[i])
        else:
            assert not g["by_path_aligner"] and all(g[key] == wg[i][key] for key in PKEYS + ("returned_reverse",)), (i, g, wg[i])
    assert 600 < n_path < 3600
