#!/usr/bin/env python
"""Reads a rocprofv3 --kernel-trace CSV of a bench.py run and says how the step's kernels sit on the hardware queues:
which queue / stream every pg_* kernel ran on, how much of the traceback + count time ran UNDER a fill kernel, and for every
step boundary whether the first fill of step n + 1 started before the count kernels (pg_fragment_kernel) of step n ended --
with the all-reduce of the counter table in the loop, that is the evidence that nothing drains the device between steps.

usage: tools/timeline_overlap.py <..._kernel_trace.csv> [out.json]"""
import collections
import csv
import json
import sys


def main(path, out_path=None):
    rows = [r for r in csv.DictReader(open(path))]
    ks = []
    for r in rows:
        n = r["Kernel_Name"]
        kind = ("fill" if "pg_fill_kernel" in n else "trace" if "pg_trace_kernel" in n else "support" if "pg_support_kernel" in n
                else "fragment" if "pg_fragment_kernel" in n else "rccl" if ("nccl" in n.lower() or "rccl" in n.lower()) else None)
        if kind:
            ks.append((kind, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Stream_Id"]))
    ks.sort(key=lambda k: k[1])
    queues = collections.defaultdict(collections.Counter)
    for k in ks:
        queues[k[0]]["queue %s / stream %s" % (k[3], k[4])] += 1
    fills = [(a, b) for kind, a, b, _, _ in ks if kind == "fill"]

    def under_fill(a, b):
        return sum(max(0, min(b, fb) - max(a, fa)) for fa, fb in fills)

    side = [(kind, a, b) for kind, a, b, _, _ in ks if kind in ("trace", "support", "fragment")]
    side_ns = sum(b - a for _, a, b in side)
    side_under = sum(under_fill(a, b) for _, a, b in side)
    frags = [(a, b) for kind, a, b, _, _ in ks if kind == "fragment"]
    boundaries = []
    for fa, fb in frags:
        nxt = [f for f in fills if f[0] > fa - 50_000_000 and f[1] > fa]  # fills still running or starting after this count began
        started_before_end = [f for f in nxt if f[0] < fb and f[1] > fb - 0]  # a fill in flight when the count kernels end
        later = [f for f in fills if f[0] >= fb]
        boundaries.append({"fragment_end_ms": fb / 1e6, "fill_in_flight_when_count_ends": bool(started_before_end),
                           "gap_to_next_fill_start_us": (min(f[0] for f in later) - fb) / 1e3 if later and not started_before_end else 0.0})
    span = (ks[-1][2] - ks[0][1]) if ks else 0
    out = {"trace": path, "kernels": dict(collections.Counter(k[0] for k in ks)), "hardware_queues": {k: dict(v) for k, v in queues.items()},
           "fill_ms_total": sum(b - a for a, b in fills) / 1e6, "side_ms_total": side_ns / 1e6,
           "side_ms_under_a_fill": side_under / 1e6, "side_fraction_under_a_fill": side_under / side_ns if side_ns else None,
           "step_boundaries": len(boundaries),
           "boundaries_with_a_fill_in_flight_when_the_count_kernels_end": sum(b["fill_in_flight_when_count_ends"] for b in boundaries),
           "rccl_kernels": sum(1 for k in ks if k[0] == "rccl"), "span_ms": span / 1e6,
           "note": "side = pg_trace_kernel + pg_support_kernel + pg_fragment_kernel (second stream); the last boundary of a timed "
                   "region is followed by the barrier, so one boundary per region has no fill in flight by construction"}
    text = json.dumps(out, indent=1)
    if out_path:
        open(out_path, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
