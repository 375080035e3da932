"""Measurement only: the path stage alone (pg_batch_path_align queued 8 times, the device drained once) on config-2 reads.
    python tools/path_probe.py [reads] [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from paragraph_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = capi.Context(0, workspace_bytes=8 << 30)
site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
graphs = ctx.upload_graphs([(site.seqs, site.edges)])
graphs.build_path_index(32)
b = ctx.new_batch()
b.upload(graphs, synth.packed_to_capi(arr))
b.path_align(fetch_flags=False)
ctx.sync()
t = time.perf_counter()
for _ in range(reps):
    b.path_align(fetch_flags=False)
ctx.sync()
s = (time.perf_counter() - t) / reps
fl = b.path_align()
print(json.dumps({"reads": n, "s_per_batch": s, "reads_per_s": n / s, "mapped_frac": float(np.mean((fl & 1) != 0)),
                  "lib": os.environ.get("PG_LIB", "production")}))
