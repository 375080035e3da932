"""CPU tests of the KmerFilter checkers (oracle/kmerfilter.py) against the reference's unit-test expectations
(src/c++/test/test_readfilter.cpp:90-166) and -- where oracle/_ref exists -- against the reference's graph-tools code."""
import random

import pytest

from oracle import counts as oc
from oracle import kmerfilter as kf
from tests import fuzzgen

DEL_NODES, DEL_EDGES = ["AGAG", "TTGG", "TTT"], [(0, 1), (1, 2), (0, 2)]
SWAP_NODES, SWAP_EDGES = ["AGAG", "T", "C", "ACAC"], [(0, 1), (0, 2), (1, 3), (2, 3)]
# (nodes, edges, k, bases, cigar, filtered, message)
VECTORS = [
    (DEL_NODES, DEL_EDGES, 3, "AGAGTT", "0[4M]1[2M]", True, "kmer_uncov_1"),
    (DEL_NODES, DEL_EDGES, 3, "AGAGTTT", "0[4M]2[3M]", False, ""),
    (SWAP_NODES, SWAP_EDGES, 4, "AGAGGACAC", "0[4M]1[1X]3[4M]", True, "kmer_uncov_1"),
    (SWAP_NODES, SWAP_EDGES, 4, "AGAGTACAC", "0[4M]1[1M]3[4M]", False, ""),
    (SWAP_NODES, SWAP_EDGES, 4, "AGAGTACAC", "0[4M]2[1X]3[4M]", True, "kmer_uncov_2"),
    (SWAP_NODES, SWAP_EDGES, 4, "AGAGCACAC", "0[4M]2[1M]3[4M]", False, ""),
]


def engines():
    out = [("port", kf.port_kmer_filter)]
    if oc.have_ref():
        out.append(("ref", kf.ref_kmer_filter))
    return out


@pytest.mark.parametrize("name,fn", engines())
def test_reference_unit_vectors(name, fn):
    for nodes, edges, k, bases, cigar, filtered, msg in VECTORS:
        kk, out = fn(nodes, edges, k, [(0, cigar, bases)])
        assert kk == k and out == [(filtered, msg)], (name, bases, cigar, out)


def rand_case(rng):
    """Graph + reads with plausible alignments: the true placement of a mutated path substring, written as a CIGAR."""
    seqs, edges = fuzzgen.rand_graph(rng, max_len=30, max_nodes=6)
    reads = []
    succ = {}
    for f, t in edges:
        succ.setdefault(f, []).append(t)
    for _ in range(10):
        node = rng.randrange(len(seqs))
        pos = rng.randrange(len(seqs[node]))
        L = rng.randint(3, 60)
        cigar, bases, start = [], [], pos
        left = rng.choice([0, 0, 0, rng.randint(1, 4)])
        while L > 0:
            take = min(L, len(seqs[node]) - pos)
            seg = seqs[node][pos:pos + take]
            ops = []
            for c in seg:
                if rng.random() < 0.06:
                    alt = rng.choice([x for x in "ACGT" if x != c])
                    bases.append(alt)
                    ops.append("X")
                else:
                    bases.append(c)
                    ops.append("M" if c != "N" else "N")
            body = ""
            i = 0
            while i < len(ops):
                j = i
                while j < len(ops) and ops[j] == ops[i]:
                    j += 1
                body += "%d%s" % (j - i, ops[i])
                i = j
            cigar.append((node, body))
            L -= take
            pos = 0
            if L > 0:
                if node not in succ:
                    break
                node = rng.choice(succ[node])
        right = rng.choice([0, 0, 0, rng.randint(1, 4)])
        b = "".join(bases)
        if left:
            b = "".join(rng.choice("ACGT") for _ in range(left)) + b
            cigar[0] = (cigar[0][0], "%dS" % left + cigar[0][1])
        if right:
            b = b + "".join(rng.choice("ACGT") for _ in range(right))
            cigar[-1] = (cigar[-1][0], cigar[-1][1] + "%dS" % right)
        reads.append((start, "".join("%d[%s]" % c for c in cigar), b))
    return seqs, edges, reads


@pytest.mark.skipif(not oc.have_ref(), reason="oracle/_ref not built")
def test_port_matches_reference_fuzz():
    rng = random.Random(31337)
    n_filtered = n_kept = 0
    for it in range(150):
        seqs, edges, reads = rand_case(rng)
        k = rng.choice([3, 4, 5, 8, 12])
        a = kf.ref_kmer_filter(seqs, edges, k, reads)
        b = kf.port_kmer_filter(seqs, edges, k, reads)
        assert a == b, (seqs, edges, k, reads, a, b)
        n_filtered += sum(1 for f, _ in a[1] if f)
        n_kept += sum(1 for f, _ in a[1] if not f)
    assert n_filtered > 100 and n_kept > 100


@pytest.mark.skipif(not oc.have_ref(), reason="oracle/_ref not built")
def test_auto_kmer_length_matches_reference():
    rng = random.Random(5)
    found = 0
    for it in range(25):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=60, max_nodes=5)
        seqs = [s if len(s) >= 12 else s + "".join(rng.choice("ACGT") for _ in range(12)) for s in seqs]
        need = rng.choice([1, 1, 2, 3])
        ka, _ = kf.ref_kmer_filter(seqs, edges, -need, [])
        kb, _ = kf.port_kmer_filter(seqs, edges, -need, [])
        assert ka == kb, (seqs, edges, need, ka, kb)
        found += ka > 0
    assert found >= 5
