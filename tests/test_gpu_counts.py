"""GPU parity of the count path: filters, per-read node/edge/sequence support and per-site counters must equal
the checker built on the reference's own graph-tools code (or the Python restatement where _ref is absent)."""
import random

import numpy as np
import pytest

from tests import fuzzgen
from tests.test_counts_oracle import ALIGNS, ALIGNS_EDGES, ALIGNS_LABELS, ALIGNS_NODES

pytestmark = pytest.mark.gpu

ALIGNS_READS = ["AAAAAAAATTTTCTTTAAAAAAAA", "TTTTTTAAAGAAAATTTTTTT", "AAAAAGCGGGGGGAAAAAA", "AAAAGCGGGGGGAAAAAA",
                "TTTTTTCCCCCCGCTTTTT", "AAAAAAAAAAAAAAAAAAA"]


def count_checker():
    from oracle import select
    return select.count_site()


def gpu_counts(ctx, graphs, labels, names, reads, gor, frag, isrev, filter_k=0, **kw):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    G.set_labels(labels, names)
    if filter_k:
        gpu_counts.filter_lengths = G.build_filter_index(filter_k)
        kw = dict(kw, use_kmer_filter=True)
    b = ctx.new_batch()
    b.upload(G, reads, gor)
    b.align(capi.AF_ALL)
    res, ops = b.download()
    b.set_fragments(frag, isrev)
    b.count(**kw)
    table, sup, path = b.download_counts()
    out = (capi.results_to_dicts(res, ops), capi.decode_supports(G, gor, sup, path, b.download_label_sets(sup)), capi.decode_counts(G, table))
    b.close()
    G.close()
    return out


def test_reference_unit_vectors_supports(gpu_ctx):
    al, sup, cnt = gpu_counts(gpu_ctx, [(ALIGNS_NODES, ALIGNS_EDGES)], [ALIGNS_LABELS], [["D", "P", "Q"]], ALIGNS_READS,
                              [0] * 6, list(range(6)), [False, True, False, False, True, False],
                              remove_nonuniq=False, use_support_filters=False)
    for i, (pos, cigar, rev, nodes, edges, labels) in enumerate(ALIGNS):
        assert al[i]["graph_pos"] == pos and al[i]["cigar"] == cigar
        assert sup[i]["status"] == 1 and sup[i]["nodes"] == nodes and sup[i]["edges"] == edges and sup[i]["labels"] == labels
    assert cnt[0]["seq_counts"] == {"P": [2, 2, 2, 0], "Q": [3, 3, 3, 0], "D": [1, 1, 1, 0]}
    assert [int(x) for x in cnt[0]["node_counts"][0]] == [6, 6, 6, 0]


def test_counts_fuzz(gpu_ctx, checker):
    from oracle import counts as oc
    check = count_checker()
    rng = random.Random(fuzzgen.salted(777))
    for kw in (dict(remove_nonuniq=True, use_support_filters=True), dict(remove_nonuniq=False, use_support_filters=True),
               dict(remove_nonuniq=False, use_support_filters=False, bad_align_frac=0.5)):
        graphs, labels, names, reads, gor, frag, isrev, want = [], [], [], [], [], [], [], []
        for gi in range(120):
            seqs, edges = fuzzgen.rand_graph(rng, max_len=40, max_nodes=6)
            lab, nm = fuzzgen.rand_labels(rng, edges)
            rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=10, max_len=90) for _ in range(rng.randint(4, 14))]
            fr = fuzzgen.rand_fragments(rng, len(rs))
            rv = [rng.random() < 0.5 for _ in rs]
            al = checker.align_batch(seqs, edges, rs)
            recs = [{"pos": a["graph_pos"], "cigar": a["cigar"], "aligned": a["score"] > 0, "unique": a["unique"],
                     "graph_reverse": rv[i] != a["returned_reverse"], "read_len": len(r), "fragment": fr[i]}
                    for i, (a, r) in enumerate(zip(al, rs))]
            want.append(check(oc.CountGraph(seqs, edges, lab, nm), recs, **kw))
            graphs.append((seqs, edges))
            labels.append(lab)
            names.append(nm)
            reads.extend(rs)
            gor.extend([gi] * len(rs))
            frag.extend(fr)
            isrev.extend(rv)
        al, sup, cnt = gpu_counts(gpu_ctx, graphs, labels, names, reads, gor, frag, isrev, **kw)
        k = 0
        for gi, w in enumerate(want):
            n = len(w["status"])
            for i in range(n):
                s = sup[k + i]
                assert s["status"] == w["status"][i], (gi, i, s, w["status"][i])
                if s["status"] == 1:
                    assert s["nodes"] == w["nodes"][i] and s["edges"] == w["edges"][i] and s["labels"] == w["labels"][i], \
                        (graphs[gi], labels[gi], reads[k + i], al[k + i], s, w["nodes"][i], w["edges"][i], w["labels"][i])
            k += n
            c = cnt[gi]
            assert not c["tallies"]["overflow"]
            assert (c["node_counts"] == w["node_counts"]).all(), (gi, c["node_counts"], w["node_counts"])
            # edge order: checker = graph edge list order; device = predecessor-CSR order -> compare by key
            cg_edges = [tuple(e) for e in graphs[gi][1]]
            for ei, e in enumerate(cg_edges):
                assert c["edge_counts"][e] == [int(x) for x in w["edge_counts"][ei]], (gi, e)
            assert c["seq_counts"] == w["seq_counts"], (gi, c["seq_counts"], w["seq_counts"])


def test_more_than_64_labels_on_a_graph(gpu_ctx, checker):
    """Label sets beyond one 64-bit word (pg_graphs_set_labels_wide: up to 256 labels on a graph; the reference's path
    families have no bound, ReadCounting.cpp:96-127): graphs with 65 .. 200 labels beside ordinary ones in ONE graph set --
    per-read status, supports and label sets, node and edge tables against the reference's code."""
    from oracle import counts as oc
    check = count_checker()
    rng = random.Random(fuzzgen.salted(6565))
    graphs, labels, names, reads, gor, frag, isrev, want = [], [], [], [], [], [], [], []
    n_wide = 0
    for gi in range(60):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=40, max_nodes=6)
        if gi % 2 == 0 and edges:
            lab, nm = fuzzgen.rand_path_labels(rng, len(seqs), edges, rng.choice([65, 66, 100, 128, 129, 200]))
            n_wide += 1
        else:
            lab, nm = fuzzgen.rand_labels(rng, edges)
        rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=10, max_len=90) for _ in range(rng.randint(4, 14))]
        fr = fuzzgen.rand_fragments(rng, len(rs))
        rv = [rng.random() < 0.5 for _ in rs]
        al = checker.align_batch(seqs, edges, rs)
        recs = [{"pos": a["graph_pos"], "cigar": a["cigar"], "aligned": a["score"] > 0, "unique": a["unique"],
                 "graph_reverse": rv[i] != a["returned_reverse"], "read_len": len(r), "fragment": fr[i]}
                for i, (a, r) in enumerate(zip(al, rs))]
        want.append(check(oc.CountGraph(seqs, edges, lab, nm), recs, remove_nonuniq=False, use_support_filters=True))
        graphs.append((seqs, edges))
        labels.append(lab)
        names.append(nm)
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        frag.extend(fr)
        isrev.extend(rv)
    assert n_wide >= 15
    al, sup, cnt = gpu_counts(gpu_ctx, graphs, labels, names, reads, gor, frag, isrev, remove_nonuniq=False, use_support_filters=True)
    k = 0
    beyond_first_word = 0
    for gi, w in enumerate(want):
        n = len(w["status"])
        for i in range(n):
            s = sup[k + i]
            assert s["status"] == w["status"][i], (gi, i, s, w["status"][i])
            if s["status"] == 1:
                assert s["nodes"] == w["nodes"][i] and s["edges"] == w["edges"][i] and s["labels"] == set(w["labels"][i]), \
                    (gi, len(names[gi]), sorted(s["labels"])[:5], sorted(w["labels"][i])[:5])
                beyond_first_word += any(names[gi].index(l) >= 64 for l in s["labels"])
        k += n
        c = cnt[gi]
        assert (c["node_counts"] == w["node_counts"]).all(), (gi, c["node_counts"], w["node_counts"])
        for ei, e in enumerate([tuple(e) for e in graphs[gi][1]]):
            assert c["edge_counts"][e] == [int(x) for x in w["edge_counts"][ei]], (gi, e)
        if len(names[gi]) <= 8:
            assert c["seq_counts"] == w["seq_counts"], (gi, c["seq_counts"], w["seq_counts"])
    assert beyond_first_word >= 20


def _rc(s):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    return "".join(comp.get(c, "N") for c in reversed(s))


@pytest.mark.parametrize("filter_k", [4, 8, -1])
def test_kmer_filter_fuzz(gpu_ctx, filter_k):
    """KmerFilter as the third filter of the chain: per-read outcome and the resulting counters."""
    from oracle import counts as oc
    from oracle import kmerfilter as kf
    from oracle import select
    check = count_checker()
    fcheck = select.kmer_filter()
    rng = random.Random(fuzzgen.salted(4040 + filter_k))
    graphs, labels, names, reads, gor, frag, isrev = [], [], [], [], [], [], []
    while len(graphs) < 80:
        seqs, edges = fuzzgen.rand_graph(rng, max_len=40, max_nodes=6)
        if filter_k < 0:
            seqs = [s + "".join(rng.choice("ACGT") for _ in range(14)) for s in seqs]
            if kf.port_kmer_filter(seqs, edges, filter_k, [])[0] < 0:
                continue
        lab, nm = fuzzgen.rand_labels(rng, edges)
        rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=10, max_len=90) for _ in range(rng.randint(4, 14))]
        gi = len(graphs)
        graphs.append((seqs, edges))
        labels.append(lab)
        names.append(nm)
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        frag.extend(fuzzgen.rand_fragments(rng, len(rs)))
        isrev.extend([rng.random() < 0.5 for _ in rs])
    kw = dict(remove_nonuniq=True, use_support_filters=True)
    al, sup0, _ = gpu_counts(gpu_ctx, graphs, labels, names, reads, gor, frag, isrev, **kw)
    al2, sup, cnt = gpu_counts(gpu_ctx, graphs, labels, names, reads, gor, frag, isrev, filter_k=filter_k, **kw)
    lengths = gpu_counts.filter_lengths
    n3 = n4 = nkeep = 0
    k0 = 0
    for gi, (seqs, edges) in enumerate(graphs):
        idx = [i for i in range(len(reads)) if gor[i] == gi]
        cand = [i for i in idx if sup0[i]["status"] == 1]
        recs = [(al[i]["graph_pos"], al[i]["cigar"], _rc(reads[i]) if al[i]["returned_reverse"] else reads[i]) for i in cand]
        kk, want = fcheck(seqs, edges, filter_k, recs)
        assert kk == lengths[gi]
        wmap = dict(zip(cand, want))
        recs2 = []
        for i in idx:
            assert al2[i]["cigar"] == al[i]["cigar"]
            if i in wmap:
                filtered, msg = wmap[i]
                exp = (1, 0) if not filtered else (2, 3 if msg == "kmer_tooshort" else 4)
                assert (sup[i]["status"], sup[i]["filter"]) == exp, (graphs[gi], reads[i], al[i], sup[i], wmap[i])
                n3 += exp[1] == 3
                n4 += exp[1] == 4
                nkeep += exp[1] == 0
            else:
                assert (sup[i]["status"], sup[i]["filter"]) == (sup0[i]["status"], sup0[i]["filter"])
            keep = sup[i]["status"] == 1
            recs2.append({"pos": al[i]["graph_pos"], "cigar": al[i]["cigar"], "aligned": al[i]["score"] > 0 and keep,
                          "unique": al[i]["unique"], "graph_reverse": isrev[i] != al[i]["returned_reverse"],
                          "read_len": len(reads[i]), "fragment": frag[i]})
        w = check(oc.CountGraph(seqs, edges, labels[gi], names[gi]), recs2, **kw)
        c = cnt[gi]
        assert (c["node_counts"] == w["node_counts"]).all(), (gi, c["node_counts"], w["node_counts"])
        assert c["seq_counts"] == w["seq_counts"]
    assert n4 > 20 and nkeep > 20
    if filter_k == 8:
        assert n3 > 0


def test_a_reused_batch_outgrows_its_count_table_without_a_wait(gpu_ctx):
    """A pooled batch object meets a graph set whose count table is larger than its last one while its own fills are still queued
    (SiteBatcher: pg_batch_align and pg_batch_count back to back under the device lock): pg_batch_count parks the old block and
    takes a new one -- no wait for the batch -- and the tables equal those of a batch object that never held anything else."""
    from paragraph_amd import capi
    rng = random.Random(fuzzgen.salted(4242))

    def plain(x):  # numpy arrays and scalars as lists and ints: comparable with ==
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        if isinstance(x, np.ndarray):
            return x.tolist()
        if isinstance(x, np.generic):
            return x.item()
        return x

    def site_set(n_sites, n_reads):
        graphs, labels, names, reads, gor, frag, isrev = [], [], [], [], [], [], []
        for gi in range(n_sites):
            graphs.append((ALIGNS_NODES, ALIGNS_EDGES))
            labels.append(ALIGNS_LABELS)
            names.append(["D", "P", "Q"])
            for r in range(n_reads):
                reads.append(ALIGNS_READS[rng.randrange(len(ALIGNS_READS))])
                gor.append(gi)
                frag.append(r // 2)
                isrev.append(bool(r & 1))
        return graphs, labels, names, reads, gor, frag, isrev

    def run(b, s):
        graphs, labels, names, reads, gor, frag, isrev = s
        G = gpu_ctx.upload_graphs(graphs)
        G.set_labels(labels, names)
        b.upload(G, reads, gor)
        b.set_fragments(frag, isrev)
        b.align(capi.AF_ALL)   # queued ...
        b.count(remove_nonuniq=True, use_support_filters=True)  # ... and counted behind it without a wait in between
        table, sup, path = b.download_counts()
        # (supports and path entries decoded: where a read's entries sit in the path array is decided by an atomic counter)
        return G, plain((capi.decode_supports(G, gor, sup, path, b.download_label_sets(sup)), capi.decode_counts(G, table)))

    small, large = site_set(3, 8), site_set(40, 12)
    reused = gpu_ctx.new_batch()
    G1, _ = run(reused, small)
    G2, got = run(reused, large)     # the count table grows from 3 sites' counters to 40 sites'
    fresh = gpu_ctx.new_batch()
    G3, want = run(fresh, large)
    assert got == want
    G4, again = run(reused, small)   # and back: the parked block was freed by the upload, the smaller table fits the larger block
    fresh_small = gpu_ctx.new_batch()
    G5, want_small = run(fresh_small, small)
    assert again == want_small
    for b in (reused, fresh, fresh_small):
        b.close()
    for G in (G1, G2, G3, G4, G5):
        G.close()
