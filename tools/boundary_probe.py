"""Measurement only: what node boundaries cost the fill kernel.  The same 200 000 config-2 reads (150 bp) are filled on graphs
with the same 502 columns cut differently: one node; the config-2 nodes as a chain (no far edge); config 2 itself (LF -> RF far
edge: seed cache + merge); a chain of 10 / 25 nodes.  One chunk per align call, the device drained between calls: fill alone.

    python tools/boundary_probe.py [reads]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paragraph_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
seq = "".join(s if isinstance(s, str) else s.decode() for s in site.seqs)
G = len(seq)


def cut(k):
    step = (G + k - 1) // k
    return [seq[i:i + step] for i in range(0, G, step)]


def chain(nodes):
    return nodes, [(i, i + 1) for i in range(len(nodes) - 1)]


cases = {
    "one_node_502": chain([seq]),
    "chain_201_100_201": chain(list(site.seqs)),
    "config2_far_edge": (list(site.seqs), list(site.edges)),
    "chain_of_10": chain(cut(10)),
    "chain_of_25": chain(cut(25)),
}
ctx = capi.Context(0, workspace_bytes=64 << 30)
packed = synth.packed_to_capi(arr)
out = {"reads": n, "columns": G}
for rep in range(2):
    for name, (nodes, edges) in cases.items():
        Gs = ctx.upload_graphs([(nodes, edges)])
        b = ctx.new_batch()
        b.upload(Gs, packed)
        b.align(capi.AF_ALL)
        ctx.sync()
        ctx.timing_enable(True)
        ctx.timing_reset()
        for _ in range(4):
            b.align(capi.AF_ALL)
            ctx.sync()
        t = ctx.timing()
        ctx.timing_enable(False)
        out.setdefault(name, []).append(t["fill_ms"] / t["fill_launches"])
        b.close()
        Gs.close()
print(json.dumps(out))
