#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6f; mkdir -p $O
cat > /tmp/pp.py <<'PY'
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from paragraph_amd import capi, synth
n = int(sys.argv[1])
ctx = capi.Context(0, workspace_bytes=8 << 30)
site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
graphs = ctx.upload_graphs([(site.seqs, site.edges)])
graphs.build_path_index(32)
b = ctx.new_batch()
b.upload(graphs, synth.packed_to_capi(arr))
b.path_align(fetch_flags=False); ctx.sync()
t = time.perf_counter()
for _ in range(8):
    b.path_align(fetch_flags=False)
ctx.sync()
s = (time.perf_counter() - t) / 8
fl = b.path_align()
print(json.dumps({"reads": n, "s_per_batch": s, "reads_per_s": n / s, "mapped_frac": float(np.mean((fl & 1) != 0)), "lib": __import__("os").environ.get("PG_LIB", "production")}))
PY
for n in 1000000 19200; do
python /tmp/pp.py $n | tee -a $O/path_probe.jsonl
PG_LIB=tools/variants/lib_pathnoatomic.so python /tmp/pp.py $n | tee -a $O/path_probe.jsonl
done
