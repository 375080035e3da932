"""Measurement only: average fill launch time in the bench's steady state (two resident 1 M-read batches alternate, the
traceback and count kernels of one chunk run on the second stream under the next chunk's fill).
    python tools/overlap_probe.py [reads] [steps] [count 0|1]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paragraph_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
count = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
ctx = capi.Context(0, workspace_bytes=64 << 30)
G = ctx.upload_graphs([(site.seqs, site.edges)])
G.set_labels([site.labels])
import numpy as np  # noqa: E402
frag = np.arange(n, dtype=np.uint32) // 2
batches = []
for _ in range(2):
    b = ctx.new_batch()
    b.upload(G, synth.packed_to_capi(arr))
    b.set_fragments(frag)
    batches.append(b)


def step(i):
    b = batches[i & 1]
    b.align(capi.AF_ALL)
    if count:
        b.count(remove_nonuniq=True, bad_align_frac=0.8)


for i in range(2):
    step(i)
ctx.sync()
ctx.timing_enable(True)
ctx.timing_reset()
import time
t0 = time.perf_counter()
for i in range(steps):
    step(i)
ctx.sync()
el = time.perf_counter() - t0
t = ctx.timing()
print(json.dumps({"reads": n, "steps": steps, "count": count, "ms_per_step": el / steps * 1e3, "reads_per_s": n * steps / el,
                  "fill_ms_per_launch": t["fill_ms"] / t["fill_launches"], "trace_ms_per_launch": t["trace_ms"] / t["trace_launches"],
                  "launches": t["fill_launches"]}))
