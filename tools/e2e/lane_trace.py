"""The lanes' timeline of one pass of the workflow (PG_WORKFLOW_TRACE): python tools/e2e/lane_trace.py [key=value ...] makes the 10 000-site
data set, runs a warm-up pass and a traced one, and prints where the wall clock of the traced pass went: set-up before the lanes start,
the first batch on the device, per phase the lane-seconds, how many lanes were in each phase over time (20 slices), the tail after
the last batch came back, and the bytes written."""
import collections
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from paragraph_amd import synth_e2e, workflow
    opts = {"threads": 16}
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        opts[k] = (v not in ("0", "false")) if k.endswith("matching") else int(v)
    d = tempfile.mkdtemp(prefix="pg_lanes_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    synth_e2e.make_dataset(d, n_sites=10000, procs=os.cpu_count() or 1)
    graphs = [l.strip() for l in open(d + "/graphs.txt") if l.strip()]
    workflow.genotype_graphs_to_file(d + "/ref.fa", d + "/manifest.txt", graphs, d + "/out.json", **opts)
    trace = d + "/trace.tsv"
    os.environ["PG_WORKFLOW_TRACE"] = trace
    t0 = time.time()
    workflow.genotype_graphs_to_file(d + "/ref.fa", d + "/manifest.txt", graphs, d + "/out.json", **opts)
    wall = time.time() - t0
    del os.environ["PG_WORKFLOW_TRACE"]
    head, rows = "", []
    for line in open(trace):
        if line.startswith("#"):
            head = line.strip()
            continue
        lane, chunk, what, a, b = line.rstrip("\n").split("\t")
        rows.append((int(lane), int(chunk), what, float(a), float(b)))
    total = max(r[4] for r in rows)
    per_phase = collections.OrderedDict()
    for _, _, what, a, b in rows:
        per_phase[what] = per_phase.get(what, 0.0) + (b - a)
    n = 20
    slices = []
    for i in range(n):
        lo, hi = total * i / n, total * (i + 1) / n
        occ = collections.Counter()
        for _, _, what, a, b in rows:
            ov = min(b, hi) - max(a, lo)
            if ov > 0:
                occ[what] += ov / (hi - lo)
        slices.append({k: round(v, 1) for k, v in occ.items()})
    first_submit = min((r[4] for r in rows if r[2] == "submit"), default=None)
    last_batch = max((r[4] for r in rows if r[2] == "batch+documents"), default=None)
    print(json.dumps({"options": opts, "wall_s": wall, "header": head, "lanes_end_s": total, "first_submit_done_s": first_submit,
                      "last_batch_back_s": last_batch, "lane_seconds_per_phase": {k: round(v, 4) for k, v in per_phase.items()},
                      "lanes_in_phase_over_time": slices, "output_bytes": os.path.getsize(d + "/out.json")}, indent=1))


if __name__ == "__main__":
    main()
