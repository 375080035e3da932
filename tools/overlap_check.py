"""Do the fills of consecutive chunks overlap on the device when they are launched on two streams?  Reads a rocprofv3 kernel trace
(…_kernel_trace.csv) and reports, for pg_fill_kernel dispatches in start order, how much of each one's duration lies under the one before it.
usage: python tools/overlap_check.py <kernel_trace.csv>"""
import csv
import json
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "pg_fill_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r.get("Stream_Id")))
rows.sort()
over = []
for (s0, e0, q0, _), (s1, e1, q1, _) in zip(rows, rows[1:]):
    over.append(max(0, min(e0, e1) - s1) / max(1, e1 - s1))
print(json.dumps({"fills": len(rows), "queues": sorted({r[2] for r in rows}), "streams": sorted({r[3] for r in rows}),
                  "mean_duration_ms": sum(e - s for s, e, _, _ in rows) / max(1, len(rows)) / 1e6,
                  "mean_fraction_under_the_previous_fill": sum(over) / max(1, len(over)),
                  "span_ms": (max(e for _, e, _, _ in rows) - min(s for s, _, _, _ in rows)) / 1e6 if rows else 0}))
