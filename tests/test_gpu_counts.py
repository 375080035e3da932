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
    from oracle import counts as oc
    if oc.have_ref():
        ref = oc.RefCounts()
        return ref.count_site
    return oc.port_count_site


def gpu_counts(ctx, graphs, labels, names, reads, gor, frag, isrev, **kw):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    G.set_labels(labels, names)
    b = ctx.new_batch()
    b.upload(G, reads, gor)
    b.align(capi.AF_ALL)
    res, ops = b.download()
    b.set_fragments(frag, isrev)
    b.count(**kw)
    table, sup, path = b.download_counts()
    out = (capi.results_to_dicts(res, ops), capi.decode_supports(G, gor, sup, path), capi.decode_counts(G, table))
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
    rng = random.Random(777)
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
