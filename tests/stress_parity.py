"""Large randomized device-vs-reference comparison (tests/test_gpu_scale.py runs two salts of it; by hand for more):

    python tests/stress_parity.py [n_graphs] [seed]

Covers the gssw stage on adversarial graphs (short nodes, N, repeats, many predecessors), bubble sites with far
predecessors (the seed-cache path), long nodes and 251..512 bp reads."""
import random
import sys
import time

sys.path.insert(0, ".")
from oracle import oracle as orc  # noqa: E402
from paragraph_amd import capi  # noqa: E402
from tests import fuzzgen  # noqa: E402
from tests.test_gpu_parity import compare, gpu_align  # noqa: E402


def run(n_graphs, seed, ctx, checker, verbose=True):
    rng = random.Random(seed)
    total = 0
    t0 = time.time()
    for block in range(0, n_graphs, 500):
        graphs, reads, gor, want = [], [], [], []
        for gi in range(min(500, n_graphs - block)):
            kind = rng.random()
            if kind < 0.6:
                seqs, edges = fuzzgen.rand_graph(rng, max_len=rng.choice([6, 40, 120]), max_nodes=rng.choice([3, 6, 9]))
                rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=5, max_len=rng.choice([60, 150, 250]))[:250] for _ in range(8)]
            elif kind < 0.8:
                seqs, edges = fuzzgen.rand_graph(rng, max_len=30, max_nodes=7, shape=rng.choice(["del", "longdel", "bubble"]))
                seqs = [s.replace("X", rng.choice("ACGTN")) for s in seqs]
                rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=20, max_len=150) for _ in range(8)]
            else:
                seqs, edges, rs = fuzzgen.long_read_case(rng, 6)
            graphs.append((seqs, edges))
            reads.extend(rs)
            gor.extend([gi] * len(rs))
            want.extend(checker.align_batch(seqs, edges, rs, cigar_stride=4096))
        got = gpu_align(ctx, graphs, reads, gor)
        compare(got, want, reads, "stress block %d" % block)
        total += len(reads)
        if verbose:
            print("block %d ok: %d reads so far, %.0fs" % (block, total, time.time() - t0), flush=True)
    return total


def main():
    n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from oracle import select
    checker = select.gssw()
    total = run(n_graphs, seed, capi.Context(0), checker)
    print("stress parity OK: %d graphs, %d reads" % (n_graphs, total))


if __name__ == "__main__":
    main()
