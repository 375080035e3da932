#!/usr/bin/env python
"""Turns the SQ counter CSVs of tools/sq_collect.sh into profiles/rNN_sq_counters.json.

usage: tools/sq_summary.py gpurun_out/sq profiles/r03_sq_counters.json [reads_in_pmc_run=200000] [kernel substring]

Per wave-step figures divide by (forward + reversed work items) x pipeline steps of the profiled launch, which the bench
line of the same command reports as roofline.trace_bytes_written_per_read: one wavefront = 4 reads x 2 strands of one graph
direction; config 2 at 150 bp: 518 steps.  SQ cycle counters are in units of 4 cycles (one wave64 issue slot)."""
import collections
import csv
import glob
import json
import os
import sys


def main(src, dst, n_reads, kernel, steps, waves_per_4_reads=2.0):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from paragraph_amd import build as pgbuild
    tot = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for path in sorted(glob.glob(os.path.join(src, "set*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
                    launches[r["Counter_Name"]].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    if not tot:
        raise SystemExit("no %s rows under %s" % (kernel, src))
    n_launch = max(len(v) for v in launches.values())
    # plain stage: a forward-graph and a reversed-graph wavefront per four reads; lean stage (kernel "pg_fill_lean_kernel<10"): a
    # reversed-graph wavefront per four reads + a forward-graph one per eight (the fourth fills of a few per cent of the reads not
    # counted: the per-wave-step figures are that much too high, the totals are what they are) = 1.5
    wave_steps = (n_reads / 4.0) * waves_per_4_reads * steps
    head_file = os.path.join(src, "head.txt")
    out = {"what": "SQ counters of the fill kernel(s) (%s) over %d config-2 reads in %d launch(es) = %.1f M wave-steps; separate "
                   "rocprofv3 --pmc passes with --kernel-trace only (tools/sq_collect.sh); cycle counters in units of 4 cycles"
                   % (kernel, n_reads, n_launch, wave_steps / 1e6),
           "kernel_source_sha": pgbuild.kernel_source_sha(),
           "collected_at_head": open(head_file).read().strip() if os.path.exists(head_file) else None,
           "reads_in_pmc_run": n_reads, "launches": n_launch, "pipeline_steps": steps, "wave_steps": wave_steps,
           "counters": dict(tot), "per_wave_step": {k: v / wave_steps for k, v in tot.items()}}
    if "SQ_INSTS_VALU" in tot and "SQ_ACTIVE_INST_VALU" in tot:
        out["cycles_per_valu_inst"] = 4.0 * tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_INSTS_VALU"]
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 200000,
         sys.argv[4] if len(sys.argv) > 4 else "pg_fill_kernel<10, false", int(sys.argv[5]) if len(sys.argv) > 5 else 518,
         float(sys.argv[6]) if len(sys.argv) > 6 else 2.0)
