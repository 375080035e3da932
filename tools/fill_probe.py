"""Measurement only: the fill and traceback kernels of one config-2 chunk on their own (one chunk per align call, so the
traceback never runs beside a fill).

    python tools/fill_probe.py [reads]

profiles/r02_fill_probe.json keeps a run of this with two temporary kernel switches (no trace stores / no profile reads
from LDS) that located the bound of pg_fill_kernel; the switches are not in the tree."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paragraph_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
ctx = capi.Context(0, workspace_bytes=64 << 30)
G = ctx.upload_graphs([(site.seqs, site.edges)])
b = ctx.new_batch()
b.upload(G, synth.packed_to_capi(arr))
b.align(capi.AF_ALL)
ctx.sync()
ctx.timing_enable(True)
ctx.timing_reset()
for _ in range(5):
    b.align(capi.AF_ALL)
ctx.sync()
t = ctx.timing()
print(json.dumps({"reads": n, "fill_ms_per_launch": t["fill_ms"] / t["fill_launches"],
                  "trace_ms_per_launch": t["trace_ms"] / t["trace_launches"], "launches": t["fill_launches"]}))
