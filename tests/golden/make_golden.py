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


if __name__ == "__main__":
    main()
