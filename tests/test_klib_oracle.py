"""CPU tests of the KlibAligner checkers (oracle/klibalign.py): the scalar ksw restatement and -- where oracle/_ref
exists -- the reference's own ksw.c, pinned on the reference's unit-test expectations and against each other."""
import random

import pytest

from oracle import klibalign as ok

# src/c++/test/test_align.cpp:38-147 (makeAlignment("klib") = default AlignmentParameters 2/-2/3/1)
LONG_REF = ("XCCTCTTAGTTCTTTGTGGAGTCTGCCTTTTCTCCCCAATCTATCCTTACCAAGTTGTCTAAGGCATGGTCCTTGCACTTATTTATACCTCTGGCTCAGACTTCT"
            "GAAGTCTGAGCTCCATACTCAGCTCAGACAGAAGTCTGAGCCCCATACTCAGCTCAGACAGAAGTCTGAGCCCCTGAGCTCCATACTCTGAT")
LONG_ALT = "TTTATACCTCTGGCTCAGACTTCTCCCCTGAGCTCCATACTCTGATACCTAACTGTTCAACTTCTCTGCATGACCATTTAATCGGCCCCCATACTGTTAT"
PAIR_VECTORS = [
    ("AAATGACGGATTG", "AAATGACCACCAGGATTG", dict(r0=0, r1=12, a0=0, a1=17, cigar="7M5I6M")),
    ("AAATGACCACCAGGATTG", "AAATGACGGATTG", dict(r0=0, r1=17, a0=0, a1=12, cigar="7M5D6M")),
    ("AAATGACGGGGCATTGCCA", "AAATGACCACCAGGATTGCCA", dict(r0=0, r1=18, a0=0, a1=20, cigar="9M3I2M1D7M")),
    (LONG_REF, LONG_ALT, dict(score=68, r0=81, r1=196, a0=0, a1=99,
                              cigar="27M2D19M3D6M2D11M1D6M2D1M1I3M1I2M2I2M1D2M1D1M1D5M7D11M")),
]

# src/c++/test/test_klibaligner.cpp:44-193
KA_NODES = ["AAAAAAAAAAA", "TTTTTTTT", "GGGGGGGG", "AAAAAAAAAAA"]
KA_PATHS = [[0, 1, 3], [0, 2, 3], [0, 3]]
KA_READS = ["AAAAAAAATTTTTTTTAAAAAAAA", "TTTTTTAAAAAAAATTTTTTT", "AAAAAGGGGGGGGAAAAAA", "AAAAGGGGGGGGAAAAAA", "TTTTTTCCCCCCCCTTTTT",
            "TTTTTTCCCCCCCCGGGGG", "GGGGGGCCCCCCCCTTTTT"]
KA_EXPECT = [(3, "0[8M]1[8M]3[8M]", 24, False), (4, "0[7M]1[8M]3[6M]", 21, True), (6, "0[5M]2[8M]3[6M]", 19, False),
             (7, "0[4M]2[8M]3[6M]", 18, False), (6, "0[5M]2[8M]3[6M]", 19, True), (0, "2[5S8M]3[6M]", 14, True),
             (6, "0[5M]2[8M6S]", 13, True)]


def engines():
    out = [("port", ok.port_klib())]
    if ok.have_ref():
        out.append(("ref", ok.ref_klib()))
    return out


@pytest.mark.parametrize("name,eng", engines())
def test_pair_vectors(name, eng):
    for ref, alt, exp in PAIR_VECTORS:
        got = eng.pair(ref, alt)
        for k, v in exp.items():
            assert got[k] == v, (name, k, got, exp)
    # KlibBasic (test_align.cpp:45-65): 2 soft-clipped, 3 matches
    got = eng.pair("AAATGACGGATTG", "TGGGA")
    assert got["a0"] == 2 and got["cigar"] == "3M" and got["a1"] == 4


@pytest.mark.parametrize("name,eng", engines())
def test_klibaligner_unit_test(name, eng):
    out = eng.align(KA_NODES, KA_PATHS, KA_READS)
    for r, (pos, cigar, score, rev) in zip(out, KA_EXPECT):
        assert r["status"] == 1 and r["mapq"] == 60 and r["unique"]
        assert (r["graph_pos"], r["cigar"], r["score"], r["is_graph_reverse"]) == (pos, cigar, score, rev), (name, r)


def _mutate(rng, s, rate):
    out = []
    i = 0
    while i < len(s):
        x = rng.random()
        if x < rate:
            out.append(rng.choice("ACGT"))
            i += 1
        elif x < rate * 1.3:
            i += rng.randint(1, 8)
        elif x < rate * 1.6:
            out.append("".join(rng.choice("ACGT") for _ in range(rng.randint(1, 8))))
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def random_case(rng, n_reads, read_len=None):
    """A small bubble graph with a few paths and reads derived from random paths (mutated, some reverse, some junk)."""
    alpha = "ACGT" if rng.random() < 0.8 else "ACGTN"
    lf = "".join(rng.choice(alpha) for _ in range(rng.randint(20, 120)))
    rf = "".join(rng.choice(alpha) for _ in range(rng.randint(20, 120)))
    n_alt = rng.randint(1, 3)
    alts = []
    for _ in range(n_alt):
        if rng.random() < 0.3 and alts:
            alts.append(_mutate(rng, alts[0], 0.1) or "A")
        else:
            alts.append("".join(rng.choice(alpha) for _ in range(rng.randint(1, 40))))
    nodes = [lf] + alts + [rf]
    last = len(nodes) - 1
    paths = [[0, i + 1, last] for i in range(n_alt)]
    if rng.random() < 0.7:
        paths.append([0, last])
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    for _ in range(n_reads):
        p = rng.choice(paths)
        seq = "".join(nodes[x] for x in p)
        L = read_len or rng.randint(10, 150)
        if rng.random() < 0.1:
            r = "".join(rng.choice("ACGT") for _ in range(L))
        else:
            a = rng.randint(0, max(0, len(seq) - 5))
            r = _mutate(rng, seq[a:a + L], rng.choice([0.0, 0.01, 0.05, 0.15]))
            if rng.random() < 0.2:
                r = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 20))) + r
            if rng.random() < 0.2:
                r = r + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 20)))
        r = r[:250] or "A"
        if rng.random() < 0.5:
            r = "".join(comp.get(c, "N") for c in reversed(r))
        reads.append(r)
    return nodes, paths, reads


@pytest.mark.skipif(not ok.have_ref(), reason="oracle/_ref not built")
def test_port_matches_reference_ksw_pairs():
    rng = random.Random(20240607)
    ref, port = ok.ref_klib(), ok.port_klib()
    for it in range(400):
        tl = rng.randint(1, 300)
        t = "".join(rng.choice("ACGTN" if it % 7 == 0 else "ACGT") for _ in range(tl))
        if rng.random() < 0.8:
            a = rng.randint(0, tl - 1)
            q = _mutate(rng, t[a:a + rng.randint(1, 200)], rng.choice([0.0, 0.02, 0.1, 0.3])) or "A"
        else:
            q = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 200)))
        params = rng.choice([(1, -4, 5, 1), (2, -2, 3, 1), (1, -1, 1, 1), (1, -4, 2, 1)])
        assert ref.pair(t, q, *params) == port.pair(t, q, *params), (t, q, params)


@pytest.mark.skipif(not ok.have_ref(), reason="oracle/_ref not built")
def test_port_matches_reference_aligner_fuzz():
    rng = random.Random(77)
    ref, port = ok.ref_klib(), ok.port_klib()
    n_bad = 0
    for it in range(40):
        nodes, paths, reads = random_case(rng, 12)
        bam = [rng.random() < 0.5 for _ in reads]
        a = ref.align(nodes, paths, reads, bam)
        b = port.align(nodes, paths, reads, bam)
        assert a == b
        n_bad += sum(1 for r in a if r["status"] == 2)
    assert n_bad > 0
