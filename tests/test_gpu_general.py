"""The general path of the gssw stage on the GPU (paragraph_amd/csrc/pg_general.hip): reads longer than the packed kernels'
512 bases and graphs longer than their 65 519 columns -- inputs the reference takes without a bound (gssw.c:527-786,
GraphAligner.cpp:110-167) -- against the reference's own gssw.c, alone and mixed with ordinary reads in one batch."""
import random

import numpy as np
import pytest

from tests import fuzzgen

pytestmark = pytest.mark.gpu


def _same(got, want):
    if want["score"] == 0:
        return got["score"] == 0 and got["status"] == 1 and got["cigar"] == "" and got["multi"] == list(want["multi"])
    return all(got[k] == want[k] for k in ("graph_pos", "score", "mapq", "cigar")) and got["unique"] == bool(want["unique"]) \
        and got["returned_reverse"] == bool(want["returned_reverse"]) and got["multi"] == list(want["multi"]) and got["status"] == 0


def _align(ctx, graphs, reads, gor):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    b = ctx.new_batch()
    b.upload(G, reads, np.asarray(gor, dtype=np.uint32))
    b.align(capi.AF_ALL)
    res, ops = b.download()
    out = capi.results_to_dicts(res, ops)
    b.close()
    G.close()
    return out


def test_long_reads_mixed_with_ordinary_ones(gpu_ctx, checker):
    """one batch over many graphs: reads of 8..1200 bases -- byte variants, wide variants and the general path side by side"""
    rng = random.Random(fuzzgen.salted(31337))
    graphs, reads, gor, want = [], [], [], []
    n_long = 0
    for g in range(60):
        if g % 3 == 0:
            seqs = [fuzzgen.rand_seq(rng, rng.randint(200, 900)) for _ in range(4)]
            edges = [(0, 1), (0, 2), (1, 2), (1, 3), (2, 3)]
        else:
            seqs, edges = fuzzgen.rand_graph(rng, max_len=rng.choice([40, 300]), max_nodes=6)
        if g % 3 == 0:
            rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=rng.choice([8, 520]), max_len=1200) for _ in range(8)]
        else:
            rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=8, max_len=rng.choice([100, 400, 1200])) for _ in range(8)]
        graphs.append((seqs, edges))
        want += checker.align_batch(seqs, edges, rs, cigar_stride=4096)
        reads += rs
        gor += [g] * len(rs)
        n_long += sum(len(r) > 512 for r in rs)
    got = _align(gpu_ctx, graphs, reads, gor)
    assert n_long > 40, n_long
    bad = [i for i, (a, w) in enumerate(zip(got, want)) if not _same(a, w)]
    assert not bad, (len(bad), bad[:5], got[bad[0]], want[bad[0]], len(reads[bad[0]]))


def test_graph_beyond_65519_columns(gpu_ctx, checker):
    """a 70 000-column insertion graph (every read on it takes the general path) beside an ordinary one"""
    rng = random.Random(fuzzgen.salted(70000))
    lf, mid, rf = fuzzgen.rand_seq(rng, 150), fuzzgen.rand_seq(rng, 70000), fuzzgen.rand_seq(rng, 150)
    wide = ([lf, mid, rf], [(0, 1), (0, 2), (1, 2)])
    hap_alt, hap_ref = lf + mid + rf, lf + rf
    reads = []
    for k in range(24):
        hap = hap_alt if k % 3 else hap_ref
        at = rng.choice([rng.randrange(0, 200), rng.randrange(len(hap) - 350, len(hap) - 150), rng.randrange(0, len(hap) - 150)])
        r = fuzzgen.mutate(rng, hap[at:at + 150], sub=0.02, indel=0.01) or "A"
        reads.append(r if k % 2 else "".join({"A": "T", "C": "G", "G": "C", "T": "A"}.get(c, "N") for c in reversed(r)))
    small = fuzzgen.rand_graph(rng, max_len=60, max_nodes=5)
    small_reads = [fuzzgen.rand_read(rng, small[0], small[1], min_len=20, max_len=150) for _ in range(8)]
    want = checker.align_batch(wide[0], wide[1], reads, threads=8) + checker.align_batch(small[0], small[1], small_reads)
    got = _align(gpu_ctx, [wide, small], reads + small_reads, [0] * len(reads) + [1] * len(small_reads))
    bad = [i for i, (a, w) in enumerate(zip(got, want)) if not _same(a, w)]
    assert not bad, (bad[:5], got[bad[0]], want[bad[0]])
    assert sum(1 for w in want[:24] if w["score"] > 100) >= 12  # (the inputs are what was meant; salts shift it: 13 seen)


def test_general_reads_are_counted_like_the_others(gpu_ctx, checker):
    """the count path takes the general path's records as they come: per-read status and supports and the per-site tables of
    graphs whose reads are 300..900 bases long equal the checker's (the reference's graph-tools code where oracle/_ref exists)"""
    from oracle import counts as oc
    from tests.test_gpu_counts import count_checker, gpu_counts
    check = count_checker()
    rng = random.Random(fuzzgen.salted(909))
    kw = dict(remove_nonuniq=True, use_support_filters=True)
    graphs, labels, names, reads, gor, frag, isrev, want = [], [], [], [], [], [], [], []
    for gi in range(24):
        seqs = [fuzzgen.rand_seq(rng, rng.randint(150, 600)) for _ in range(4)]
        edges = [(0, 1), (0, 2), (1, 2), (1, 3), (2, 3)]
        lab, nm = fuzzgen.rand_labels(rng, edges)
        rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=300, max_len=900) for _ in range(6)]
        fr = fuzzgen.rand_fragments(rng, len(rs))
        rv = [rng.random() < 0.5 for _ in rs]
        al = checker.align_batch(seqs, edges, rs, cigar_stride=4096)
        recs = [{"pos": a["graph_pos"], "cigar": a["cigar"], "aligned": a["score"] > 0, "unique": a["unique"],
                 "graph_reverse": rv[i] != a["returned_reverse"], "read_len": len(r), "fragment": fr[i]} for i, (a, r) in enumerate(zip(al, rs))]
        want.append(check(oc.CountGraph(seqs, edges, lab, nm), recs, **kw))
        graphs.append((seqs, edges))
        labels.append(lab)
        names.append(nm)
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        frag.extend(fr)
        isrev.extend(rv)
    assert sum(len(r) > 512 for r in reads) > 12
    al, sup, cnt = gpu_counts(gpu_ctx, graphs, labels, names, reads, gor, frag, isrev, **kw)
    k = 0
    for gi, w in enumerate(want):
        n = len(w["status"])
        for i in range(n):
            s = sup[k + i]
            assert s["status"] == w["status"][i], (gi, i, len(reads[k + i]), s, w["status"][i], al[k + i])
            if s["status"] == 1:
                assert s["nodes"] == w["nodes"][i] and s["edges"] == w["edges"][i] and s["labels"] == w["labels"][i], (gi, i)
        k += n
        c = cnt[gi]
        assert (c["node_counts"] == w["node_counts"]).all(), (gi, c["node_counts"], w["node_counts"])
        for ei, e in enumerate(graphs[gi][1]):
            assert c["edge_counts"][tuple(e)] == [int(x) for x in w["edge_counts"][ei]], (gi, e)
        assert c["seq_counts"] == w["seq_counts"], (gi, c["seq_counts"], w["seq_counts"])


def test_graph_beyond_4095_nodes_and_runs_beyond_4095_bases(gpu_ctx, checker):
    """a 5 000-node chain with skip edges (node ids need more than the 12 bits the packed kernels' column words hold) and reads
    of 4 500 .. 5 200 bases whose match runs are longer than one CIGAR element holds (PG_OP_MAX_LEN: the pieces print as one)"""
    rng = random.Random(fuzzgen.salted(5000))
    seqs = [fuzzgen.rand_seq(rng, rng.randint(1, 6)) for _ in range(5000)]
    edges = [(i, i + 1) for i in range(4999)] + [(i, i + 2) for i in range(0, 4998, 37)] + [(i, i + 5) for i in range(3, 4990, 211)]
    path = "".join(seqs)
    reads = []
    for k in range(28):
        L = rng.choice([40, 150, 150, 320, 600])
        at = rng.randrange(0, len(path) - L)
        r = fuzzgen.mutate(rng, path[at:at + L], sub=rng.choice([0.0, 0.02]), indel=rng.choice([0.0, 0.01])) or "A"
        reads.append(r if k % 2 else "".join({"A": "T", "C": "G", "G": "C", "T": "A"}.get(c, "N") for c in reversed(r)))
    long_nodes = [fuzzgen.rand_seq(rng, 6000), fuzzgen.rand_seq(rng, 300), fuzzgen.rand_seq(rng, 400)]
    long_edges = [(0, 1), (0, 2), (1, 2)]
    long_reads = [long_nodes[0][500:5500], long_nodes[0][900:6000] + long_nodes[2][:100],
                  fuzzgen.mutate(rng, long_nodes[0][100:4700], sub=0.0005, indel=0.0)]
    want = (checker.align_batch(seqs, edges, reads, threads=8, cigar_stride=8192)
            + checker.align_batch(long_nodes, long_edges, long_reads, threads=3, cigar_stride=8192))
    got = _align(gpu_ctx, [(seqs, edges), (long_nodes, long_edges)], reads + long_reads, [0] * len(reads) + [1] * len(long_reads))
    bad = [i for i, (a, w) in enumerate(zip(got, want)) if not _same(a, w)]
    assert not bad, (bad[:5], got[bad[0]], want[bad[0]])
    import re
    assert any(int(n) > 4095 for w in want[:28] for n in re.findall(r"(\d+)\[", w["cigar"]))
    assert any(int(m) > 4095 for w in want[28:] for m in re.findall(r"(\d+)M", w["cigar"]))


def test_path_stage_takes_a_read_beyond_4095_bases(gpu_ctx):
    """PathAligner (exact matching, default ON in `paragraph`) on a 5 000-base read that matches a node run exactly: one
    CIGAR element per node, the 4 600-base run inside the long node in pieces that print as one (PathAligner.cpp:121-161)."""
    from paragraph_amd import capi
    rng = random.Random(fuzzgen.salted(4600))
    nodes = [fuzzgen.rand_seq(rng, 300), fuzzgen.rand_seq(rng, 6000), fuzzgen.rand_seq(rng, 500)]
    edges = [(0, 1), (0, 2), (1, 2)]
    read = nodes[0][200:] + nodes[1][:4600]
    short = nodes[0][100:250]
    G = gpu_ctx.upload_graphs([(nodes, edges)])
    G.build_path_index(32)
    b = gpu_ctx.new_batch()
    b.upload(G, [read, short])
    b.path_align()
    res, ops = b.download()
    got = capi.results_to_dicts(res, ops)
    from tests.test_gpu_path import PKEYS, path_checker
    want = path_checker()(nodes, edges, [read, short], 32)
    assert want[0]["status"] == 1 and want[0]["cigar"] == "0[100M]1[4600M]" and want[0]["score"] == len(read), want[0]
    for g, w in zip(got, want):
        assert g["by_path_aligner"] == bool(w["status"]), (g, w)
        if w["status"]:
            assert all(g[key] == w[key] for key in PKEYS) and g["returned_reverse"] == w["is_graph_reverse"], (g, w)
    b.close()
    G.close()
