import sys, os, json, numpy as np
sys.path.insert(0, "/root/repo")
from paragraph_amd import capi, synth
n_sites=int(sys.argv[1])
sites = synth.mixed_sites(n_sites, seed=11)
graphs = [(s.site.seqs, s.site.edges) for s in sites]
reads = np.concatenate([s.reads for s in sites])
gor = np.concatenate([np.full(len(s.reads), i, dtype=np.uint32) for i, s in enumerate(sites)])
L = reads.shape[1]
off = (np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
ctx = capi.Context(0, workspace_bytes=8 << 30)
G = ctx.upload_graphs(graphs)
b = ctx.new_batch(); b.upload(G, (off, reads.tobytes()), gor)
for mode in (0, 2):
    ctx.set_lean(mode)
    ctx.timing_enable(True); ctx.timing_reset()
    b.align(); ctx.sync()
    t = ctx.timing()
    print(mode, {k: t[k] for k in ("fill_launches", "fill_ms", "lean_fused_launches", "lean_fused_ms", "fills")})
