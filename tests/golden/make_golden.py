#!/usr/bin/env python
"""Generates tests/golden/*.json from the REFERENCE's own gssw.c (oracle/_ref/libpg_ref.so, built by
oracle/Makefile from /root/reference/external/gssw/gssw.c).  Run in the build container only:

    python tests/golden/make_golden.py

Each fixture is data only: graphs, reads and the outputs the reference arithmetic produced for them
(graph_pos, CIGAR, score, MAPQ, uniqueness, strand, the four multi flags and the four fill scores).
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.oracle import RefOracle  # noqa: E402
from paragraph_amd import synth  # noqa: E402
from tests import fuzzgen  # noqa: E402


def dump(name, sets, note):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "source": "reference gssw.c via oracle/ref_harness.c",
                   "note": note, "sets": sets}, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes")


def main():
    R = RefOracle()
    # 1. randomized adversarial graphs (short nodes, repeats, N, indels, lower case)
    sets = []
    for seqs, edges, reads in fuzzgen.cases(4242, 60, 8):
        sets.append({"nodes": seqs, "edges": edges, "reads": reads, "expected": R.align_batch(seqs, edges, reads)})
    dump("fuzz_small.json", sets, "fuzzgen.cases(4242, 60, 8)")
    # 2. longer nodes / reads up to 250 bp
    rng = random.Random(99)
    sets = []
    for _ in range(12):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=160, max_nodes=5)
        reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=90, max_len=250)[:250] for _ in range(6)]
        sets.append({"nodes": seqs, "edges": edges, "reads": reads, "expected": R.align_batch(seqs, edges, reads)})
    dump("fuzz_long.json", sets, "rand_graph(max_len=160,max_nodes=5) x12, reads 90-250 bp")
    # 3. BASELINE configs[1] sample: DEL graph 201/100/201, 150 bp reads, seed 2
    site, reads = synth.config2_reads(256, read_len=150, seed=2)
    sets = [{"nodes": site.seqs, "edges": site.edges, "reads": reads,
             "expected": R.align_batch(site.seqs, site.edges, reads)}]
    dump("config2_sample.json", sets, "synth.config2_reads(256, read_len=150, seed=2)")
    # 4. insertion and long-deletion templates
    contig = synth.random_contig(11, 4000)
    ins = synth.random_contig(12, 120).decode()
    s1 = synth.ins_site(contig, 1500, ins, flank=150)
    s2 = synth.longdel_site(contig, 1000, 2500, flank=150)
    sets = []
    for s, seed in ((s1, 21), (s2, 22)):
        reads = synth.simulate_reads(s, 96, 150, seed, n_frac=0.002)
        sets.append({"nodes": s.seqs, "edges": s.edges, "reads": reads,
                     "expected": R.align_batch(s.seqs, s.edges, reads)})
    dump("templates_ins_longdel.json", sets, "ins_site / longdel_site on random_contig(11,4000), 96 reads each")
    # 5. reads of 251..512 bp: gssw's 16-bit word mode (incl. GraphAligner's byte-pointer multi-node scan)
    rng = random.Random(251)
    sets = []
    for _ in range(10):
        seqs, edges, reads = fuzzgen.long_read_case(rng, 6)
        sets.append({"nodes": seqs, "edges": edges, "reads": reads,
                     "expected": R.align_batch(seqs, edges, reads, cigar_stride=2048)})
    dump("fuzz_word.json", sets, "fuzzgen.long_read_case(Random(251), 6) x10: reads 251-512 bp")
    stage_fixtures()


def dump_stage(name, source, note, sets):
    os.makedirs(os.path.join(HERE, "stages"), exist_ok=True)
    path = os.path.join(HERE, "stages", name)
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "source": source, "note": note, "sets": sets}, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes")


def stage_fixtures():
    """Fixtures of the optional cascade stages and the KmerFilter, from the reference-backed checkers (oracle/_ref)."""
    from oracle import klibalign, kmerfilter, pathalign
    from tests.test_klib_oracle import random_case
    from tests.test_kmerfilter_oracle import rand_case as filter_case
    # klib stage: the reference's ksw.c under the restated KlibAligner wrapper
    rng = random.Random(606)
    K = klibalign.ref_klib()
    sets = []
    for _ in range(25):
        nodes, paths, reads = random_case(rng, 8)
        sets.append({"nodes": nodes, "paths": paths, "reads": reads, "expected": K.align(nodes, paths, reads)})
    dump_stage("klib.json", "reference ksw.c via oracle/ref_harness.c + klib_glue.h", "test_klib_oracle.random_case(Random(606), 8) x25", sets)
    # path stage: graph-tools extendPathMatching etc. via oracle/ref_counts.cpp
    rng = random.Random(707)
    sets = []
    for _ in range(25):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=80, max_nodes=6)
        seqs = [s.replace("X", "A") for s in seqs]
        reads = []
        for _ in range(8):
            p = fuzzgen.rand_path_seq(rng, seqs, edges)
            L = rng.randint(20, 90)
            st = rng.randrange(max(1, len(p) - 10))
            r = p[st:st + L] or "A"
            if rng.random() < 0.3:
                r = fuzzgen.mutate(rng, r, sub=0.02, indel=0.0) or "A"
            if rng.random() < 0.4:
                r = pathalign._rc(r)
            reads.append(r)
        k = rng.choice([12, 16, 32])
        sets.append({"nodes": seqs, "edges": edges, "k": k, "reads": reads, "expected": pathalign.ref_path_align(seqs, edges, reads, k)})
    dump_stage("path.json", "reference graph-tools via oracle/ref_counts.cpp", "rand_graph(Random(707)) x25, 8 reads each, k in {12,16,32}", sets)
    # KmerFilter
    rng = random.Random(808)
    sets = []
    for _ in range(40):
        seqs, edges, reads = filter_case(rng)
        k = rng.choice([3, 4, 5, 8, 12])
        kk, out = kmerfilter.ref_kmer_filter(seqs, edges, k, reads)
        sets.append({"nodes": seqs, "edges": edges, "k": k, "reads": [list(r) for r in reads], "expected": [[bool(f), m] for f, m in out]})
    dump_stage("kmerfilter.json", "reference graph-tools via oracle/ref_counts.cpp", "test_kmerfilter_oracle.rand_case(Random(808)) x40", sets)


if __name__ == "__main__":
    main()
