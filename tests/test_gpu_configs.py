"""GPU parity on the other BASELINE.json configurations (small samples the oracle finishes in seconds):
config 3 = mixed DEL / long-DEL / INS sites with 30x paired reads (alignments AND counts),
config 5 = long ALT nodes (kb-sized inline sequences) with 250 bp reads."""
import numpy as np
import pytest

from tests import fuzzgen

pytestmark = pytest.mark.gpu

KEYS = ("graph_pos", "score", "mapq", "unique", "returned_reverse", "multi", "cigar")


def _align_and_count(ctx, sites, remove_nonuniq=True):
    from paragraph_amd import capi
    graphs = [(s.site.seqs, s.site.edges) for s in sites]
    G = ctx.upload_graphs(graphs)
    G.set_labels([s.site.labels for s in sites])
    reads = np.concatenate([s.reads for s in sites])
    gor = np.concatenate([np.full(len(s.reads), i, dtype=np.uint32) for i, s in enumerate(sites)])
    frag = np.concatenate([s.fragment for s in sites])
    rev = np.concatenate([s.is_reverse for s in sites])
    L = reads.shape[1]
    off = (np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
    b = ctx.new_batch()
    b.upload(G, (off, reads.tobytes()), gor)
    b.set_fragments(frag, rev)
    b.align(capi.AF_ALL)
    res, ops = b.download()
    b.count(remove_nonuniq=remove_nonuniq)
    table, sup, path = b.download_counts()
    out = (capi.results_to_dicts(res, ops), capi.decode_supports(G, gor, sup, path), capi.decode_counts(G, table))
    b.close()
    G.close()
    return out


def test_config3_mixed_sites_sample(gpu_ctx, checker):
    from oracle import counts as oc
    from paragraph_amd import synth
    from tests.test_gpu_counts import count_checker
    check = count_checker()
    sites = synth.mixed_sites(40, seed=3)
    al, sup, cnt = _align_and_count(gpu_ctx, sites)
    k = 0
    kinds = set()
    for si, s in enumerate(sites):
        kinds.add(s.site.kind)
        reads = [row.tobytes().decode() for row in s.reads]
        want = checker.align_batch(s.site.seqs, s.site.edges, reads, threads=8)
        for i, w in enumerate(want):
            g = al[k + i]
            if w["score"] == 0:
                assert g["status"] == 1
            else:
                assert all(g[key] == w[key] for key in KEYS if key != "multi") and fuzzgen.multi_equal(g, w), (si, i, reads[i], g, w)
        recs = [{"pos": w["graph_pos"], "cigar": w["cigar"], "aligned": w["score"] > 0, "unique": w["unique"],
                 "graph_reverse": bool(s.is_reverse[i]) != w["returned_reverse"], "read_len": len(reads[i]),
                 "fragment": int(s.fragment[i])} for i, w in enumerate(want)]
        labels = sorted({l for v in s.site.labels.values() for l in v})
        wc = check(oc.CountGraph(s.site.seqs, s.site.edges, s.site.labels, labels), recs, remove_nonuniq=True)
        for i in range(len(reads)):
            assert sup[k + i]["status"] == wc["status"][i], (si, i)
            if wc["status"][i] == 1:
                assert sup[k + i]["nodes"] == wc["nodes"][i] and sup[k + i]["edges"] == wc["edges"][i] \
                    and sup[k + i]["labels"] == wc["labels"][i], (si, i, al[k + i], sup[k + i])
        c = cnt[si]
        assert (c["node_counts"] == wc["node_counts"]).all(), (si, s.site.kind)
        for ei, e in enumerate(s.site.edges):
            assert c["edge_counts"][tuple(e)] == [int(x) for x in wc["edge_counts"][ei]], (si, e)
        assert c["seq_counts"] == wc["seq_counts"], (si, c["seq_counts"], wc["seq_counts"])
        k += len(reads)
    assert kinds == {"del", "longdel", "ins"}
    assert k > 5000


def test_config5_long_nodes_250bp(gpu_ctx, checker):
    from paragraph_amd import synth
    from tests.test_gpu_parity import compare, gpu_align
    graphs, reads, gor, want = [], [], [], []
    for gi, alt_len in enumerate((2100, 4000, 8000)):
        site = synth.long_node_site(50 + gi, alt_len)
        rs = synth.simulate_reads(site, 48, 250, 60 + gi, indel_frac=0.05, random_frac=0.02)
        graphs.append((site.seqs, site.edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(checker.align_batch(site.seqs, site.edges, rs, threads=8))
    got = gpu_align(gpu_ctx, graphs, reads, gor)
    compare(got, want, reads, "config5")
