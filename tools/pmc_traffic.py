#!/usr/bin/env python
"""Turns the rocprofv3 PMC CSVs collected by tools/pmc_collect.sh into profiles/traffic_rNN.json.

usage: tools/pmc_traffic.py gpurun_out/pmc_r01 profiles/traffic_r01.json [reads_in_pmc_run=200000]

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; FETCH_SIZE
under-reports wide coalesced reads by 2x on gfx950 -- the factor is MEASURED here from the calibration run
(copy of 1 GiB) instead of assumed; WRITE_SIZE is calibrated on zero_/copy_ of 1 GiB."""
import collections
import csv
import json
import os
import sys

GIB = float(1 << 30)


def load(path):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return agg


def main(src, dst, n_reads):
    cal_f = load(os.path.join(src, "cal_FETCH_SIZE", "cal_counter_collection.csv"))
    cal_w = load(os.path.join(src, "cal_WRITE_SIZE", "cal_counter_collection.csv"))
    copy_f = [v for k, v in cal_f.items() if "copyBuffer" in k][0]
    copy_w = [v for k, v in cal_w.items() if "copyBuffer" in k][0]
    zero_w = [v for k, v in cal_w.items() if "elementwise" in k][0]
    fetch_factor = GIB / (sum(copy_f) / len(copy_f))
    write_factor = GIB / ((sum(copy_w) / len(copy_w) + sum(zero_w) / len(zero_w)) / 2)
    ben_f = load(os.path.join(src, "bench_FETCH_SIZE", "bench_counter_collection.csv"))
    ben_w = load(os.path.join(src, "bench_WRITE_SIZE", "bench_counter_collection.csv"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from paragraph_amd import build as pgbuild
    head_file = os.path.join(src, "head.txt")  # written by tools/pmc_collect.sh on the box (the snapshot has no .git)
    out = {"source": src, "reads_in_pmc_run": n_reads, "kernel_source_sha": pgbuild.kernel_source_sha(),
           "collected_at_head": open(head_file).read().strip() if os.path.exists(head_file) else None,
           "calibration": {"fetch_factor": fetch_factor, "write_factor": write_factor,
                           "how": "torch copy_ / zero_ of 1 GiB under the same --pmc pass"},
           "kernels": {}}
    for k in sorted(set(ben_f) | set(ben_w)):
        if "pg_" not in k:
            continue
        f = sum(ben_f.get(k, [])) * fetch_factor
        w = sum(ben_w.get(k, [])) * write_factor
        n = max(len(ben_f.get(k, [])), len(ben_w.get(k, [])))
        out["kernels"][k] = {"launches": n, "fetch_bytes": f, "write_bytes": w,
                             "hbm_bytes_per_read": (f + w) / n_reads, "hbm_bytes_per_launch": (f + w) / max(n, 1)}
    # the fill kernels of the run together: pg_fill_kernel (plain stage), or the lean stage's pg_fill_lean_kernel<C, 2> (reversed-graph
    # fills) + <C, 3> (forward-graph fills of the instance items; the second, small launch of the fourth fills included), or its fused
    # form pg_fill_lean_fused_kernel<C> (+ <C, 3> for the second, small launch)
    fills = {k: v for k, v in out["kernels"].items() if "pg_fill_kernel" in k or "pg_fill_lean" in k}
    out["fill_kernels"] = sorted(fills)
    out["hbm_bytes_per_read"] = sum(v["hbm_bytes_per_read"] for v in fills.values())
    out["hbm_bytes_per_launch"] = sum(v["fetch_bytes"] + v["write_bytes"] for v in fills.values()) / max(1, max(v["launches"] for v in fills.values()))
    out["fill_fetch_bytes_per_read"] = sum(v["fetch_bytes"] for v in fills.values()) / n_reads
    out["fill_write_bytes_per_read"] = sum(v["write_bytes"] for v in fills.values()) / n_reads
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 200000)
