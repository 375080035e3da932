"""BASELINE configs[4]-style probe: INV/DUP-like graphs with 2-8 kb ALT nodes, 250 bp reads (gssw stage + counts).
Usage: python tools/config5_probe.py [n_sites] [reads_per_site]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from paragraph_amd import capi, synth  # noqa: E402


def main():
    n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    per_site = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    ctx = capi.Context(0, workspace_bytes=64 << 30)
    graphs, reads, gor, cells, b_alg = [], [], [], 0, 0
    L = 250
    for gi in range(n_sites):
        alt = 2000 + (gi * 6007) % 6001
        site = synth.long_node_site(100 + gi, alt)
        rs = synth.simulate_reads(site, per_site, L, 500 + gi, indel_frac=0.01, random_frac=0.005)
        graphs.append((site.seqs, site.edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        G = sum(len(s) for s in site.seqs)
        cells += 4 * L * G * len(rs)
        b_alg += (6 * L * G + L + 64) * len(rs)
    G_ = ctx.upload_graphs(graphs)
    b = ctx.new_batch()
    b.upload(G_, reads, gor)
    b.align(capi.AF_ALL)
    ctx.sync()
    t = time.perf_counter()
    reps = 2
    for _ in range(reps):
        b.align(capi.AF_ALL)
    ctx.sync()
    s = (time.perf_counter() - t) / reps
    n = len(reads)
    print(json.dumps({"sites": n_sites, "reads": n, "read_len": L, "mean_graph_len": cells / (4 * L * n), "s_per_batch": s,
                      "reads_per_s": n / s, "gcups": cells / s / 1e9, "alg_bytes_per_read": b_alg / n,
                      "alg_GBps": b_alg / s / 1e9, "frac_of_8TBps": b_alg / s / 8e12}))


if __name__ == "__main__":
    main()
