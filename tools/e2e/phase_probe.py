"""Where a pass of the workflow spends its lanes' time, per phase of SiteBatcher::run (PG_BATCH_TIMING=1 prints every batch's phases on
stderr): python tools/e2e/phase_probe.py <n_sites> [key=value ...]   e.g. path_sequence_matching=1"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, time, json
sys.path.insert(0, %r)
from paragraph_amd import workflow
d, opts = sys.argv[1], json.loads(sys.argv[2])
graphs = [l.strip() for l in open(d + "/graphs.txt") if l.strip()]
workflow.genotype_graphs_to_file(d + "/ref.fa", d + "/manifest.txt", graphs, d + "/out.json", **opts)   # warm-up
print("MARK", file=sys.stderr, flush=True)
t0 = time.time()
workflow.genotype_graphs_to_file(d + "/ref.fa", d + "/manifest.txt", graphs, d + "/out.json", **opts)
print(json.dumps({"seconds": time.time() - t0, "sites": len(graphs)}))
"""


def main():
    from paragraph_amd import synth_e2e
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    opts = {"threads": 16}
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        opts[k] = (v not in ("0", "false")) if k.endswith("matching") else int(v)
    d = tempfile.mkdtemp(prefix="pg_phase_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    synth_e2e.make_dataset(d, n_sites=n, procs=os.cpu_count() or 1)
    env = dict(os.environ, PG_BATCH_TIMING="1")
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT, d, json.dumps(opts)], env=env, capture_output=True, text=True)
    err = p.stderr.split("MARK", 1)[-1]
    phases = collections.OrderedDict()
    for m in re.finditer(r"\[SiteBatcher\]\s+(.+?)\s+([0-9.]+) ms", err):
        phases.setdefault(m.group(1), [0.0, 0])
        phases[m.group(1)][0] += float(m.group(2))
        phases[m.group(1)][1] += 1
    out = json.loads(p.stdout.strip().splitlines()[-1]) if p.stdout.strip() else {"error": p.stderr[-2000:]}
    import resource
    ru = resource.getrusage(resource.RUSAGE_CHILDREN)
    out["cpu_s_both_passes_and_startup"] = ru.ru_utime + ru.ru_stime
    out["options"] = opts
    out["phase_lane_ms"] = {k: round(v[0], 1) for k, v in phases.items()}
    out["batches"] = max((v[1] for v in phases.values()), default=0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
