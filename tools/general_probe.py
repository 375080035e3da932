"""Measurement only: the general path of the gssw stage (reads beyond 512 bases, graphs beyond 65 519 columns) on its own.

    python tools/general_probe.py            (PG_LIB=<variant> for A/B runs)

Three cases: 2 000 reads of 600 bases and 500 reads of 2 000 bases on a config-2-like graph (nodes 1 200 / 600 / 1 200), 260 reads of
150 bases on a 70 000-column insertion graph.  Prints reads/s and DP cell updates/s of each."""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from paragraph_amd import capi  # noqa: E402


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, s, sub=0.01):
    return "".join(rng.choice("ACGT") if rng.random() < sub else c for c in s)


def run(ctx, graph, reads):
    G = ctx.upload_graphs([graph])
    b = ctx.new_batch()
    b.upload(G, reads, np.zeros(len(reads), dtype=np.uint32))
    b.align(capi.AF_ALL)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(2):
        b.align(capi.AF_ALL)
    ctx.sync()
    s = (time.perf_counter() - t) / 2
    res, _ = b.download()
    b.close()
    G.close()
    cols = sum(len(x) for x in graph[0])
    cells = 4.0 * sum(len(r) for r in reads) * cols
    return {"reads": len(reads), "read_len": len(reads[0]), "graph_len": cols, "s_per_batch": round(s, 4), "reads_per_s": round(len(reads) / s),
            "gcups": round(cells / s / 1e9, 2), "aligned": int((res["score"] > 0).sum())}


def main():
    rng = random.Random(5)
    ctx = capi.Context(0, workspace_bytes=16 << 30)
    lf, mid, rf = rand_seq(rng, 1200), rand_seq(rng, 600), rand_seq(rng, 1200)
    g2 = ([lf, mid, rf], [(0, 1), (0, 2), (1, 2)])
    out = {}
    for name, n, L in (("600bp", 2000, 600), ("2000bp", 500, 2000)):
        hap = lf + mid + rf
        reads = []
        for _ in range(n):
            at = rng.randrange(0, len(hap) - L)
            reads.append(mutate(rng, hap[at:at + L]))
        out[name] = run(ctx, g2, reads)
    big = rand_seq(rng, 70000)
    gw = ([lf[:150], big, rf[:150]], [(0, 1), (0, 2), (1, 2)])
    hap = lf[:150] + big + rf[:150]
    reads = []
    for _ in range(260):
        at = rng.randrange(0, len(hap) - 150)
        reads.append(mutate(rng, hap[at:at + 150]))
    out["150bp_on_70k_columns"] = run(ctx, gw, reads)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
