"""CPU tests of the count-path checkers: the pure-Python restatement (oracle/counts.py) against the
reference's unit-test expectations and -- where oracle/_ref exists -- against the reference's own
graph-tools code (oracle/_ref/libpg_refcounts.so) on randomized alignments."""
import os
import random

import pytest

from oracle import counts as oc
from tests import fuzzgen

ALIGNS_NODES = ["AAAAAAAAAAA", "TTTTTTTT", "GGGGGGGG", "AAAAAAAAAAA"]
ALIGNS_EDGES = [(0, 1), (0, 2), (0, 3), (1, 3), (2, 3)]
ALIGNS_LABELS = {(0, 1): ["P"], (1, 3): ["P"], (0, 2): ["Q"], (2, 3): ["Q"], (0, 3): ["D"]}
# graphPos, graphCigar, reverse strand and the supports of src/c++/test/test_paragraph_parts.cpp:111-143
ALIGNS = [(3, "0[8M]1[4M1X3M]3[8M]", False, {0, 1, 3}, {(0, 1), (1, 3)}, {"P"}),
          (4, "0[7M]1[4M1X3M]3[6M]", True, {0, 1, 3}, {(0, 1), (1, 3)}, {"P"}),
          (6, "0[5M]2[1M1X6M]3[6M]", False, {0, 2, 3}, {(0, 2), (2, 3)}, {"Q"}),
          (7, "0[4M]2[1M1X6M]3[6M]", False, {0, 2, 3}, {(0, 2), (2, 3)}, {"Q"}),
          (6, "0[5M]2[1M1X6M]3[6M]", True, {0, 2, 3}, {(0, 2), (2, 3)}, {"Q"}),
          (0, "0[11M]3[8M]", False, {0, 3}, {(0, 3)}, {"D"})]
ALIGNS_LENS = [24, 21, 19, 18, 19, 19]


def aligns_records():
    return [{"pos": p, "cigar": c, "aligned": True, "unique": True, "graph_reverse": rev, "read_len": L, "fragment": i}
            for i, ((p, c, rev, _, _, _), L) in enumerate(zip(ALIGNS, ALIGNS_LENS))]


def test_port_supports_match_reference_unit_test():
    g = oc.CountGraph(ALIGNS_NODES, ALIGNS_EDGES, ALIGNS_LABELS)
    out = oc.port_count_site(g, aligns_records(), remove_nonuniq=False, use_support_filters=False)
    for i, (_, _, _, nodes, edges, labels) in enumerate(ALIGNS):
        assert out["nodes"][i] == nodes and out["edges"][i] == edges and out["labels"][i] == labels
    assert out["seq_counts"] == {"P": [2, 2, 1, 1], "Q": [3, 3, 2, 1], "D": [1, 1, 1, 0]}
    assert [int(x) for x in out["node_counts"][0]] == [6, 6, 4, 2]


def test_port_disambiguation_unit_test():
    """src/c++/test/test_disambiguation.cpp:97-105: R / R / none / D."""
    nodes = ["AAAAAAAAAA", "TTTTTTTTTT", "TTTTTTTTTT", "GGGGGGGGGG", "AAAAAAAAAA"]
    edges = [(0, 1), (0, 4), (1, 2), (1, 3), (2, 4), (3, 4)]
    labels = {(0, 1): ["R"], (1, 2): ["R"], (2, 4): ["R"], (0, 4): ["D"]}
    g = oc.CountGraph(nodes, edges, labels)
    cig = ["0[10M]1[10M]2[10M]4[10M]", "0[10M]1[10M]2[1M]", "0[10M]1[10M]3[10M]4[10M]", "0[10M]4[10M]"]
    recs = [{"pos": 0, "cigar": c, "aligned": True, "unique": True, "graph_reverse": False, "read_len": L,
             "fragment": i} for i, (c, L) in enumerate(zip(cig, [40, 21, 40, 20]))]
    out = oc.port_count_site(g, recs, remove_nonuniq=False, use_support_filters=False)
    assert [sorted(s) for s in out["labels"]] == [["R"], ["R"], [], ["D"]]


def test_bad_align_rounding():
    g = oc.CountGraph(["A" * 200], [], {})
    # L = 150: round(0.8 * 150) = 120 -> 119 aligned is bad, 120 is fine; L = 18: round(14.4) = 14
    def rec(clip, L):
        return {"pos": 0, "cigar": "0[%dS%dM]" % (clip, L - clip), "aligned": True, "unique": True,
                "graph_reverse": False, "read_len": L, "fragment": 0}
    out = oc.port_count_site(g, [rec(31, 150), rec(30, 150), rec(4, 18), rec(5, 18)], remove_nonuniq=False)
    assert out["status"] == [2, 1, 1, 2]


def make_case(rng, checker):
    seqs, edges = fuzzgen.rand_graph(rng, max_len=40, max_nodes=6)
    labels, names = fuzzgen.rand_labels(rng, edges)
    reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=10, max_len=90) for _ in range(rng.randint(4, 14))]
    al = checker.align_batch(seqs, edges, reads)
    frag = fuzzgen.rand_fragments(rng, len(reads))
    isrev = [rng.random() < 0.5 for _ in reads]
    recs = [{"pos": a["graph_pos"], "cigar": a["cigar"], "aligned": a["score"] > 0, "unique": a["unique"],
             "graph_reverse": isrev[i] != a["returned_reverse"], "read_len": len(r), "fragment": frag[i]}
            for i, (a, r) in enumerate(zip(al, reads))]
    return oc.CountGraph(seqs, edges, labels, names), recs, reads, frag, isrev


def same(x, y):
    return (x["status"] == y["status"] and x["nodes"] == y["nodes"] and x["edges"] == y["edges"]
            and x["labels"] == y["labels"] and (x["node_counts"] == y["node_counts"]).all()
            and (x["edge_counts"] == y["edge_counts"]).all() and x["seq_counts"] == y["seq_counts"])


def test_port_vs_reference_graphtools_randomized(checker):
    if not oc.have_ref():
        pytest.skip("oracle/_ref/libpg_refcounts.so not built")
    ref = oc.RefCounts()
    rng = random.Random(31337)
    n = 0
    for _ in range(400):
        g, recs, _, _, _ = make_case(rng, checker)
        for kw in (dict(remove_nonuniq=True, use_support_filters=True), dict(remove_nonuniq=False, use_support_filters=True),
                   dict(remove_nonuniq=False, use_support_filters=False, bad_align_frac=0.5)):
            x = ref.count_site(g, recs, **kw)
            y = oc.port_count_site(g, recs, **kw)
            assert x["rc"] == 0 and y["rc"] == 0
            assert same(x, y), (g.nodes, g.edges, g.edge_labels, recs, kw)
            n += 1
    assert n == 1200


def test_checkers_with_more_than_64_labels(checker):
    """Label sets beyond one 64-bit word: the checker built on the reference's own graph-tools / Disambiguation code (whose label
    sets are std::sets of names: no bound) hands them out as four words; it must agree with the restatement on graphs of
    65 .. 200 labels."""
    if not oc.have_ref():
        pytest.skip("oracle/_ref/libpg_refcounts.so not built")
    ref = oc.RefCounts()
    rng = random.Random(6565)
    n = beyond = 0
    for _ in range(60):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=40, max_nodes=6)
        if not edges:
            continue
        lab, names = fuzzgen.rand_path_labels(rng, len(seqs), edges, rng.choice([65, 66, 100, 128, 129, 200]))
        reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=10, max_len=90) for _ in range(rng.randint(4, 12))]
        fr = fuzzgen.rand_fragments(rng, len(reads))
        rv = [rng.random() < 0.5 for _ in reads]
        al = checker.align_batch(seqs, edges, reads)
        recs = [{"pos": a["graph_pos"], "cigar": a["cigar"], "aligned": a["score"] > 0, "unique": a["unique"],
                 "graph_reverse": rv[i] != a["returned_reverse"], "read_len": len(r), "fragment": fr[i]}
                for i, (a, r) in enumerate(zip(al, reads))]
        g = oc.CountGraph(seqs, edges, lab, names)
        x = ref.count_site(g, recs, remove_nonuniq=False)
        y = oc.port_count_site(g, recs, remove_nonuniq=False)
        assert x["rc"] == 0 and same(x, y), (seqs, edges, recs)
        beyond += sum(1 for ls in x["labels"] if any(names.index(l) >= 64 for l in ls))
        n += 1
    assert n >= 40 and beyond >= 50


def test_path_checker_unit_vectors_and_port_vs_ref():
    """PathAligner checkers: reference unit tests (src/c++/test/test_pathaligner.cpp:37-144, k = 16) and the
    pure-Python port against the harness built on the reference's graph-tools."""
    from oracle import pathalign as pa
    g1 = (["AAAAAAAAA", "CCCC", "GGGGGGGGG"], [(0, 1), (0, 2), (1, 2)])
    reads = ["AAAAAAAAGGGGGGGG", "CCCCCCCCTTTTTTTT", "AAAAAAAACCCCGGGG", "CCCCGGGGTTTTTTTT", "AAAAAAAAGGGGGGGGG",
             "CCCCCCCCCTTTTTTTTT"]
    want = [(1, "0[8M]2[8M]", 16, False), (1, "0[8M]2[8M]", 16, True), (1, "0[8M]1[4M]2[4M]", 16, False),
            (1, "0[8M]1[4M]2[4M]", 16, True), (1, "0[8M]2[9M]", 17, False), (0, "0[9M]2[9M]", 18, True)]
    got = pa.port_path_align(*g1, reads, 16)
    for g, (pos, cigar, score, rev) in zip(got, want):
        assert (g["status"], g["graph_pos"], g["cigar"], g["score"], g["is_graph_reverse"], g["mapq"]) == \
            (1, pos, cigar, score, rev, 60)
    g2 = (["GGGGGGGGGGGG", "CCCCCCCCCCCCCCCC", "GGGGGGGGGGGGGTGGG"], [(0, 1), (0, 2), (1, 2)])
    m = pa.port_path_align(*g2, ["CCCCCCCCCCCCGGGGGGGGGGGG"], 16)[0]
    assert (m["graph_pos"], m["cigar"], m["score"], m["unique"], m["mapq"]) == (4, "1[12M]2[12M]", 24, False, 0)
    if not pa.have_ref():
        return
    rng = random.Random(5)
    n = 0
    for _ in range(300):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=60, max_nodes=6)
        k = rng.choice([8, 12, 16, 32])
        rs = []
        for _ in range(8):
            p = fuzzgen.rand_path_seq(rng, seqs, edges)
            st = rng.randrange(max(1, len(p)))
            r = p[st:st + rng.randint(k, 80)]
            if rng.random() < 0.3:
                r = fuzzgen.mutate(rng, r, sub=0.02, indel=0.0)
            if rng.random() < 0.4:
                r = pa._rc(r)
            rs.append(r or "A")
        assert pa.ref_path_align(seqs, edges, rs, k) == pa.port_path_align(seqs, edges, rs, k), (seqs, edges, rs, k)
        n += len(rs)
    assert n == 2400


def test_kmer_checker_unit_vectors():
    """KmerAlignerTest.Aligns, src/c++/test/test_kmeraligner.cpp:149-193 (KmerAligner<10>)."""
    from oracle import kmeralign as ka
    nodes = ["AAAAAAAAAAA", "TTTTTTTT", "GGGGGGGG", "AAAAAAAAAAA"]
    paths = [[0, 1, 3], [0, 2, 3], [0, 3]]
    reads = ["AAAAAAAATTTTTTTTAAAAAAAA", "TTTTTTAAAAAAAATTTTTTT", "AAAAAGGGGGGGGAAAAAA", "AAAAGGGGGGGGAAAAAA",
             "TTTTTTCCCCCCCCTTTTT", "AAAAAAAAAAAAAAAAAAA"]
    want = [(1, 3, "0[8M]1[8M]3[8M]", 24, False, 60), (1, 4, "0[7M]1[8M]3[6M]", 21, True, 60),
            (1, 6, "0[5M]2[8M]3[6M]", 19, False, 60), (1, 7, "0[4M]2[8M]3[6M]", 18, False, 60),
            (1, 6, "0[5M]2[8M]3[6M]", 19, True, 60), (2, 0, "0[11M]3[8M]", 19, False, 0)]
    got = ka.port_kmer_align(nodes, paths, reads, 10)
    for g, (st, pos, cigar, score, rev, mapq) in zip(got, want):
        assert (g["status"], g["graph_pos"], g["cigar"], g["score"], g["is_graph_reverse"], g["mapq"]) == \
            (st, pos, cigar, score, rev, mapq)


def test_fragment_graph_length_equals_the_references(tmp_path):
    """The graph length of a read pair -- what `fragment_statistics` is made of -- as the host library computes it
    (paragraph::fragmentStatistics over the restated graphtools::GraphCoordinates) against common::Fragment::addRead's
    arithmetic on the reference's OWN GraphCoordinates.cpp (compiled from the graph-tools tarball): random graphs, the
    alignments the reference's gssw.c gives two reads on them."""
    import json
    import random
    import subprocess
    from oracle import counts as oc
    from oracle import oracle as orc
    from paragraph_amd import build
    from tests import fuzzgen
    if not (oc.have_ref() and orc.have_ref()):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    exe = build.build_genotyping_test()
    exe = os.path.join(os.path.dirname(exe), "test_hostio")
    ref, gssw = oc.RefCounts(), orc.RefOracle()
    rng = random.Random(20260927)
    fasta = tmp_path / "none.fa"
    fasta.write_text(">x\nACGT\n")
    (tmp_path / "none.fa.fai").write_text("x\t4\t3\t4\t5\n")
    checked = with_gap = no_path = 0
    for it in range(60):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=rng.choice([8, 40, 90]), max_nodes=8)
        seqs = [s.replace("X", "A") for s in seqs]
        reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=6, max_len=60) for _ in range(12)]
        al = [a for a in gssw.align_batch(seqs, edges, reads) if a["score"] > 0 and a["cigar"]]
        pairs = [(a["graph_pos"], a["cigar"], b["graph_pos"], b["cigar"]) for a, b in zip(al[0::2], al[1::2])]
        if not pairs:
            continue
        want = ref.pair_lengths(oc.CountGraph(seqs, edges), pairs)
        graph_json = tmp_path / ("g%d.json" % it)
        graph_json.write_text(json.dumps({"nodes": [{"name": "n%d" % i, "sequence": s} for i, s in enumerate(seqs)],
                                          "edges": [{"from": "n%d" % a, "to": "n%d" % b} for a, b in edges],
                                          "target_regions": ["x:1-2"]}))
        p = subprocess.run([exe, "--pair-lengths", str(graph_json), str(fasta)], input="".join("%d %s %d %s\n" % q for q in pairs),
                           capture_output=True, text=True, timeout=60)
        assert p.returncode == 0, p.stderr
        # (lengths go through the statistics' doubles; an alignment that ends on the first base of its last node has an
        # "unknown" end in the reference -- all ones -- and the fragment length it then computes wraps around: kept as it is)
        got = [None if l == "-" else float(l) for l in p.stdout.split()]
        want_f = [None if w is None else float(w) for w in want]
        assert got == want_f, (seqs, edges, [(q, g, w) for q, g, w in zip(pairs, got, want) if g != (None if w is None else float(w))][:3])
        checked += len(pairs)
        no_path += sum(w is None for w in want)
        with_gap += sum(1 for q, w in zip(pairs, want) if w is not None and q[1].split("[")[0] != q[3].split("[")[0])
    assert checked > 150 and with_gap > 40, (checked, with_gap, no_path)


def test_alignment_statistics_equal_the_references_summarize_alignments(tmp_path):
    """`alignment_statistics` (per node / edge / allele: reads by strand, match-base depth, mismatch / gap / clip rates,
    average score, contig length) as the host library builds it against the reference's OWN summarizeAlignments +
    AlignmentStatistics.cpp (compiled as they lie: oracle/_ref/libpg_refstats.so) on random labelled graphs, the alignments the
    reference's gssw.c gives the reads and the sequence supports its disambiguation gives them."""
    import json
    import math
    import random
    import subprocess
    from oracle import counts as oc
    from oracle import oracle as orc
    from oracle import stats as ost
    from paragraph_amd import build
    from tests import fuzzgen
    if not (oc.have_ref() and orc.have_ref() and ost.have_ref()):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    exe = os.path.join(os.path.dirname(build.build_genotyping_test()), "test_hostio")
    counts, gssw = oc.RefCounts(), orc.RefOracle()
    rng = random.Random(4711)
    fasta = tmp_path / "none.fa"
    fasta.write_text(">x\nACGT\n")
    (tmp_path / "none.fa.fai").write_text("x\t4\t3\t4\t5\n")

    def same(a, b, where):
        if isinstance(a, dict):
            assert isinstance(b, dict) and set(a) == set(b), (where, sorted(a), sorted(b) if isinstance(b, dict) else b)
            for k in a:
                same(a[k], b[k], where + "/" + k)
        elif isinstance(a, float) or isinstance(b, float):
            assert math.isclose(a, b, rel_tol=1e-12, abs_tol=1e-15), (where, a, b)
        else:
            assert a == b, (where, a, b)

    n_reads = n_elements = 0
    for it in range(90):
        shape = rng.choice([None, None, "longdel", "bubble"])
        seqs, edges = fuzzgen.rand_graph(rng, max_len=rng.choice([10, 40]), max_nodes=7, shape=shape)
        terminals = it % 3 == 0 and len(seqs) >= 3
        names = ["n%d" % i for i in range(len(seqs))]
        if terminals:  # source / sink: one-base "X" nodes once loaded (GraphInput.cpp:86-89), left out of the statistics
            names[0], names[-1] = "source", "sink"
            seqs = ["X"] + list(seqs[1:-1]) + ["X"]
        labels, label_names = fuzzgen.rand_labels(rng, edges)
        reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=8, max_len=70) for _ in range(16)]
        al = gssw.align_batch(seqs, edges, reads)
        recs = [{"pos": a["graph_pos"], "cigar": a["cigar"], "aligned": a["score"] > 0, "unique": a["unique"], "graph_reverse": rng.random() < 0.5,
                 "read_len": len(r), "fragment": i} for i, (a, r) in enumerate(zip(al, reads))]
        sup = counts.count_site(oc.CountGraph(seqs, edges, labels, label_names), recs, remove_nonuniq=False, use_support_filters=True)
        mapped = [{"pos": rec["pos"], "cigar": rec["cigar"], "reverse": rec["graph_reverse"], "score": a["score"], "sequences": sorted(sup["labels"][i])}
                  for i, (rec, a) in enumerate(zip(recs, al)) if sup["status"][i] == 1]
        if not mapped:
            continue
        want = ost.alignment_statistics(names, seqs, edges, labels, mapped)
        graph_json = tmp_path / ("s%d.json" % it)
        graph_json.write_text(json.dumps({
            "nodes": [{"name": nm, "sequence": s} for nm, s in zip(names, seqs)],
            "edges": [dict({"from": names[a], "to": names[b]}, **({"sequences": labels[(a, b)]} if labels.get((a, b)) else {})) for a, b in edges],
            "sequencenames": label_names, "target_regions": ["x:1-2"]}))
        lines = "".join("%d %s %d %d %s\n" % (m["pos"], m["cigar"], 1 if m["reverse"] else 0, m["score"], ",".join(m["sequences"]) or "-") for m in mapped)
        p = subprocess.run([exe, "--alignment-statistics", str(graph_json), str(fasta)], input=lines, capture_output=True, text=True, timeout=60)
        assert p.returncode == 0, p.stderr
        got = json.loads(p.stdout)
        same(want, got, "graph %d" % it)
        n_reads += len(mapped)
        n_elements += sum(len(want[k]) for k in ("nodes", "edges", "alleles"))
    assert n_reads > 250 and n_elements > 300, (n_reads, n_elements)
