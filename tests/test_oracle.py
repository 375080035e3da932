"""CPU tests (no GPU): the oracle against the reference's known-answer vectors, the committed golden
fixtures (generated from the reference's own gssw.c) and -- when oracle/_ref is present -- the real
gssw.c on fresh randomized inputs."""
import glob
import json
import os
import subprocess

import pytest

from tests import fuzzgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.json")))
KEYS = ("graph_pos", "score", "mapq", "unique", "returned_reverse", "multi", "scores", "cigar")


@pytest.fixture(scope="module")
def port():
    from oracle import oracle as orc
    if not orc.have_port():
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True)
    return orc.PortOracle()


def test_port_reference_unit_vectors(port):
    """ParagraphTest.Aligns, src/c++/test/test_paragraph_parts.cpp:46-159."""
    from tests.test_gpu_parity import ALIGNS_EXPECTED, ALIGNS_GRAPH, ALIGNS_READS
    got = port.align_batch(*ALIGNS_GRAPH, ALIGNS_READS)
    for g, (pos, cigar, score, mapq, rev) in zip(got, ALIGNS_EXPECTED):
        assert (g["graph_pos"], g["cigar"], g["score"], g["mapq"], g["returned_reverse"]) == (pos, cigar, score, mapq, rev)
        assert g["unique"] is True


def test_disambiguation_unit_vectors(port):
    """Graph and reads of src/c++/test/test_disambiguation.cpp:44-95: alignments must exist and be
    consistent between threads=1 and threads=4 (the reference's AlignsMultithreaded idea)."""
    nodes = ["AAAAAAAAAAA", "TTTTTTTT", "GGGGGGGG", "AAAAAAAAAAA"]
    edges = [(0, 1), (0, 2), (0, 3), (1, 3), (2, 3)]
    reads = ["AAAAAAAATTTTTTTTAAAAAAAA", "AAAAAAAAGGGGGGGGAAAAAAAA", "TTTTTTTTTTTTCCCCCCCCTTTT", "AAAAAAAAAAAAAAAAAA"]
    a = port.align_batch(nodes, edges, reads, threads=1)
    b = port.align_batch(nodes, edges, reads, threads=4)
    assert a == b
    assert a[0]["cigar"] == "0[8M]1[8M]3[8M]" and a[1]["cigar"] == "0[8M]2[8M]3[8M]"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_port_matches_golden(port, path):
    with open(path) as f:
        fx = json.load(f)
    n = 0
    for s in fx["sets"]:
        got = port.align_batch(s["nodes"], [tuple(e) for e in s["edges"]], s["reads"])
        for g, w, r in zip(got, s["expected"], s["reads"]):
            for k in KEYS:
                assert g[k] == w[k], (path, r, k, g, w)
            n += 1
    assert n > 0


def test_port_vs_reference_gssw_randomized(port):
    """Fresh randomized differential test against the reference's own gssw.c (only where oracle/_ref
    exists, i.e. in the build container and on boxes that received the prebuilt .so)."""
    from oracle import oracle as orc
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    ref = orc.RefOracle()
    n = 0
    for seed, kw in ((101, dict(max_len=40)), (102, dict(max_len=6)), (103, dict(max_len=120, max_nodes=5))):
        for seqs, edges, reads in fuzzgen.cases(seed, 250, 8, **kw):
            a = ref.align_batch(seqs, edges, reads)
            b = port.align_batch(seqs, edges, reads)
            assert a == b, (seqs, edges)
            n += len(reads)
    assert n == 6000


def test_port_vs_reference_gssw_word_mode(port):
    """Reads of 251..512 bp: gssw restarts the fill in its 16-bit word mode once a score reaches 255 - bias
    (gssw.c:380, 527-786); the port models that as plain arithmetic.  Pinned here against the real gssw.c,
    with reads close enough to the graph that most of them do overflow the byte mode."""
    import random
    from oracle import oracle as orc
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    ref = orc.RefOracle()
    rng = random.Random(2511)
    n = n_over = 0
    n_edge = n_multi = 0
    for it in range(150):
        if it % 3 == 0:
            seqs, edges = fuzzgen.rand_graph(rng, max_len=260, max_nodes=5)
            reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=251, max_len=512)[:512] for _ in range(6)]
        else:
            seqs, edges, reads = fuzzgen.long_read_case(rng)
        a = ref.align_batch(seqs, edges, reads)
        b = port.align_batch(seqs, edges, reads)
        assert a == b, (seqs, edges)
        n += len(reads)
        n_over += sum(1 for x in a if x["score"] >= 251)
        n_edge += sum(1 for x in a if 251 <= max(x["scores"]) <= 255)
        n_multi += sum(1 for x in a if max(x["scores"]) >= 251 and any(x["multi"]))
    assert n == 900 and n_over > 300 and n_edge > 20 and n_multi > 0


def test_fill_level_outputs_vs_reference(port):
    from oracle import oracle as orc
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    ref = orc.RefOracle()
    for seqs, edges, reads in fuzzgen.cases(7, 150, 4, max_len=60):
        g, h = ref.graph(seqs, edges), port.graph(seqs, edges)
        for r in reads:
            for d in (0, 1):
                x, y = g.fill(d, r.upper()), h.fill(d, r.upper())
                if x["score"] == 0:
                    x["max_node"] = y["max_node"] = 0  # stale gssw max_node on all-zero fills: not modelled
                assert x == y, (seqs, edges, r, d)
        g.close()
        h.close()


def test_degenerate_reads(port):
    nodes, edges = ["ACGTACGT", "TTTT", "GGGGCCCC"], [(0, 1), (0, 2), (1, 2)]
    out = port.align_batch(nodes, edges, ["NNNNNN", "n", "ACGU"])
    assert out[0]["score"] == 0 and out[0]["cigar"] == "" and out[0]["graph_pos"] == 0 and out[0]["mapq"] == 0
    assert out[1]["score"] == 0
    # 'U' scores as 'A' (gssw.c:4213-4216) but is printed as a mismatch character comparison
    assert out[2]["score"] >= 3
