"""pg_batch_retire_exact_matches: the exact shortcut in front of the gssw stage (include/paragraph_amd.h).  A read the path stage
matches over its whole length exactly ONCE (and, when the match is on the reverse complement, whose forward strand holds a k-mer
that is in no path of the graph) keeps the record GraphAligner::alignRead(AF_ALL) would have written
(src/c++/lib/grm/GraphAligner.cpp:308-404) and skips its four fills.  Checked against the plain gssw stage of the same library --
itself compared with the reference's gssw.c in tests/test_gpu_parity.py -- on every field the reference's Read carries."""
import random

import numpy as np
import pytest

from tests import fuzzgen

pytestmark = pytest.mark.gpu

KEYS = ("graph_pos", "score", "mapq", "unique", "returned_reverse", "cigar", "clipped", "status", "by_path_aligner")


def plain(ctx, graphs, reads, gor):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    b = ctx.new_batch()
    b.upload(G, reads, gor)
    b.align(capi.AF_ALL)
    res, ops = b.download()
    out = capi.results_to_dicts(res, ops)
    b.close()
    G.close()
    return out


def shortcut(ctx, graphs, reads, gor, k):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    G.build_path_index(k)
    b = ctx.new_batch()
    b.upload(G, reads, gor)
    flags = b.path_align()
    b.retire_exact_matches()
    b.align(capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS)
    res, ops = b.download()
    out = capi.results_to_dicts(res, ops)
    b.close()
    G.close()
    return flags, out


def compare(want, got, reads, graphs, gor):
    """-> number of reads that skipped their fills (the other strand's score is unknown for them: -1)"""
    skipped = 0
    for i, (w, g) in enumerate(zip(want, got)):
        where = (i, reads[i], graphs[gor[i]] if gor is not None else None, w, g)
        assert all(w[key] == g[key] for key in KEYS), where
        if -1 in g["strand_score"]:
            skipped += 1
            s = 1 if g["returned_reverse"] else 0
            # what is known of the forced record: the returned strand scored the read's length, none of its flags is set
            assert g["score"] == len(reads[i]) and g["strand_score"][s] == w["strand_score"][s] == len(reads[i]), where
            assert g["unique"] and g["mapq"] == 60 and w["multi"][s] == 0 and w["multi"][2 + s] == 0 and g["multi"] == [0, 0, 0, 0], where
            if s == 1:
                assert w["strand_score"][0] < len(reads[i]), where
        else:
            assert fuzzgen.multi_equal_known(w, g) and w["strand_score"] == g["strand_score"], where
    return skipped


def test_config2_reads(gpu_ctx):
    from paragraph_amd import synth
    site, reads = synth.config2_reads(20000, read_len=150, seed=5)
    graphs = [(site.seqs, site.edges)]
    want = plain(gpu_ctx, graphs, reads, None)
    flags, got = shortcut(gpu_ctx, graphs, reads, None, 32)
    skipped = compare(want, got, reads, graphs, [0] * len(reads))
    assert 0.15 * len(reads) < skipped <= int(np.count_nonzero(flags & 1))  # (0.99 ** 150 = 22 % of the reads are exact)


def _reads_of(rng, seqs, edges, k, n):
    from oracle import pathalign as pa
    out = []
    for _ in range(n):
        p = fuzzgen.rand_path_seq(rng, seqs, edges)
        L = rng.randint(k, 120)
        st = rng.randrange(max(1, len(p)))
        r = p[st:st + L]
        u = rng.random()
        if u < 0.25:
            r = fuzzgen.mutate(rng, r, sub=0.02, indel=0.01)
        elif u < 0.30 and len(r) > 4:
            r = r[:len(r) // 2] + pa._rc(r[:len(r) // 2])  # a palindrome: both strands spell the same
        elif u < 0.35 and r:
            j = rng.randrange(len(r))
            r = r[:j] + "N" + r[j + 1:]
        if rng.random() < 0.5:
            r = pa._rc(r)
        if rng.random() < 0.05:
            r = r.lower()
        out.append(r or "A")
    return out


@pytest.mark.parametrize("k", [8, 16, 32])
def test_fuzz(gpu_ctx, k):
    """graphs made to break the argument: bubbles whose alleles are near copies (several walks spell a read), repeats (homopolymer
    and periodic nodes: k-mers with many paths), one-base and N nodes, palindromic reads, N and lower case in reads, short k"""
    rng = random.Random(fuzzgen.salted(4200 + k))
    graphs, reads, gor = [], [], []
    for gi in range(400):
        u = rng.random()
        if u < 0.3:
            a = fuzzgen.rand_seq(rng, rng.randint(k // 2, 70))
            b = a if rng.random() < 0.3 else (fuzzgen.mutate(rng, a, sub=0.03, indel=0.0) or "A")
            lf, rf = fuzzgen.rand_seq(rng, rng.randint(1, 80)), fuzzgen.rand_seq(rng, rng.randint(1, 80))
            seqs, edges = [lf, a, b, rf], [(0, 1), (0, 2), (1, 3), (2, 3)] + ([(0, 3)] if rng.random() < 0.5 else [])
        elif u < 0.4:
            unit = fuzzgen.rand_seq(rng, rng.randint(k, 60))
            seqs, edges = [fuzzgen.rand_seq(rng, 30), unit, unit, fuzzgen.rand_seq(rng, 30)], [(0, 1), (1, 2), (0, 2), (2, 3), (1, 3)]
        else:
            seqs, edges = fuzzgen.rand_graph(rng, max_len=rng.choice([6, 40, 90]), max_nodes=rng.choice([3, 6, 9]))
        rs = _reads_of(rng, seqs, edges, k, 10)
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
    want = plain(gpu_ctx, graphs, reads, gor)
    flags, got = shortcut(gpu_ctx, graphs, reads, gor, k)
    skipped = compare(want, got, reads, graphs, gor)
    assert skipped > 300


def test_needs_the_path_stage(gpu_ctx):
    from paragraph_amd import capi, synth
    site, reads = synth.config2_reads(64, read_len=150, seed=1)
    G = gpu_ctx.upload_graphs([(site.seqs, site.edges)])
    b = gpu_ctx.new_batch()
    b.upload(G, reads)
    with pytest.raises(capi.PgError):
        b.retire_exact_matches()
    b.close()
    G.close()
