#!/usr/bin/env python
"""Device time of a WORKFLOW-sized batch (192 mixed sites, ~19 000 paired 150 bp reads, as a lane of the BAM -> genotypes job uploads it)
through the plain gssw stage and through the lean one, alone on the device: two batch objects aligned in turn, ms per pg_batch_align +
pg_batch_count.  usage: tools/small_batch_probe.py [sites=192] [rounds=200] [workspace GiB=64: the host workflow's default; a chunk is at
most half of it]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from paragraph_amd import capi, synth  # noqa: E402


def main(n_sites, rounds, ws_gib):
    sites = synth.mixed_sites(n_sites, seed=11)
    graphs = [(s.site.seqs, s.site.edges) for s in sites]
    reads = np.concatenate([s.reads for s in sites])
    gor = np.concatenate([np.full(len(s.reads), i, dtype=np.uint32) for i, s in enumerate(sites)])
    L = reads.shape[1]
    off = (np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
    out = {"sites": n_sites, "reads": int(len(reads)), "workspace_gib": ws_gib}
    ctx = capi.Context(0, workspace_bytes=int(ws_gib * (1 << 30)))
    G = ctx.upload_graphs(graphs)
    G.set_labels([s.site.labels for s in sites])
    bs = [ctx.new_batch() for _ in range(4)]
    for b in bs:
        b.upload(G, (off, reads.tobytes()), gor)
        b.set_fragments(np.concatenate([s.fragment for s in sites]), np.concatenate([s.is_reverse for s in sites]))
    for name, mode in (("plain", 0), ("lean_every_chunk", 2), ("plain_again", 0), ("lean_again", 2)):
        ctx.set_lean(mode)
        for b in bs:
            b.align()
            b.count()
        ctx.sync()
        t0 = time.perf_counter()
        for r in range(rounds):
            b = bs[r % len(bs)]
            b.align()
            b.count()
        ctx.sync()
        out[name] = {"ms_per_batch": (time.perf_counter() - t0) / rounds * 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 192, int(sys.argv[2]) if len(sys.argv) > 2 else 200, float(sys.argv[3]) if len(sys.argv) > 3 else 64.0)
