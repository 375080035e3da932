"""Ad-hoc GPU debugging helper (not a test): prints the first mismatches against the oracle."""
import sys, time
sys.path.insert(0, ".")
from paragraph_amd import capi, synth
from oracle import oracle as orc
from tests import fuzzgen
from tests.test_gpu_parity import ALIGNS_GRAPH, ALIGNS_READS, gpu_align, KEYS

from oracle import select
chk = select.gssw()
ctx = capi.Context(0)

def report(name, got, want, reads, limit=5):
    bad = 0
    for i, (g, w) in enumerate(zip(got, want)):
        if any(g[k] != w[k] for k in KEYS):
            bad += 1
            if bad <= limit:
                print("MISMATCH", name, i, reads[i]); print("  gpu ", g); print("  want", w)
    print(name, ":", len(reads), "reads,", bad, "mismatches")

got = gpu_align(ctx, [ALIGNS_GRAPH], ALIGNS_READS)
want = chk.align_batch(*ALIGNS_GRAPH, ALIGNS_READS)
report("aligns", got, want, ALIGNS_READS, 6)

graphs, reads, gor, want = [], [], [], []
for gi, (seqs, edges, rs) in enumerate(fuzzgen.cases(2024, 300, 10)):
    graphs.append((seqs, edges)); reads.extend(rs); gor.extend([gi]*len(rs)); want.extend(chk.align_batch(seqs, edges, rs))
got = gpu_align(ctx, graphs, reads, gor)
report("fuzz", got, want, reads)

site, reads = synth.config2_reads(4096, read_len=150, seed=2)
want = chk.align_batch(site.seqs, site.edges, reads, threads=8)
t0 = time.time(); got = gpu_align(ctx, [(site.seqs, site.edges)], reads); print("gpu time", time.time()-t0)
report("config2", got, want, reads)
