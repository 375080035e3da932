"""Throughput of the optional cascade stages (path / k-mer / klib) and of the gssw stage on config-2 reads.
Usage: python tools/stage_probe.py [n_reads]   (prints one JSON object)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from paragraph_amd import capi, synth  # noqa: E402


def timed(ctx, fn, reps=3):
    fn()
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    ctx = capi.Context(0, workspace_bytes=64 << 30)
    site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
    graphs = ctx.upload_graphs([(site.seqs, site.edges)])
    paths = [[[0, 1, 2], [0, 2]]]
    graphs.build_path_index(32)
    graphs.build_kmer_index(paths, 16)
    graphs.build_klib_index(paths)
    b = ctx.new_batch()
    b.upload(graphs, synth.packed_to_capi(arr))
    out = {"reads": n, "read_len": 150, "graph_len": int(site.total_len)}
    for name, fn in (("path", b.path_align), ("kmer", b.kmer_align), ("klib", b.klib_align), ("gssw", lambda: b.align(capi.AF_ALL))):
        s = timed(ctx, fn)
        flags = None
        if name != "gssw":
            flags = fn()
        out[name] = {"s_per_batch": s, "reads_per_s": n / s,
                     "mapped_frac": None if flags is None else float(np.mean((flags & 1) != 0))}
    assert graphs.klib_error() == 0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
