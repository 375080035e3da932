#!/usr/bin/env python
"""gpurun_out/stage_counters (tools/stage_counters.sh) -> profiles/rNN_stage_counters.json: per kernel the LAST launch's duration,
HBM bytes (FETCH_SIZE / WRITE_SIZE x the calibration factors of the same collection, MI355X_MICROARCH.md's HBM section), SQ
counters, and what bounds it: the share of the 8 TB/s peak its bytes make over its duration, the share of the 1 024 SIMDs' VALU
issue slots its instructions hold (4 cycles per wave64 instruction, the clock of profiles/r04_clock_probe.json), its LDS
instructions -- or neither (latency / occupancy: few wavefronts, dependent loads).

usage: tools/stage_counters_summary.py gpurun_out/stage_counters profiles/r05_stage_counters.json [n_reads]"""
import collections
import csv
import glob
import json
import os
import sys

GIB = float(1 << 30)
HBM_PEAK = 8.0e12
SIMDS = 1024


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def last_values(path):
    """{kernel: {counter: value of the kernel's last dispatch}} (a counter CSV lists one row per dispatch and counter)"""
    by = collections.defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            by[(short(r["Kernel_Name"]), r["Counter_Name"])][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    out = collections.defaultdict(dict)
    for (k, c), d in by.items():
        out[k][c] = d[max(d)]
    return out


def find(src, sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main(src, dst, n_reads):
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, root)
    clock_ghz = 2.387
    try:
        with open(os.path.join(root, "profiles", "r04_clock_probe.json")) as f:
            clock_ghz = json.load(f)["under_fill_load"]["xcd_clock_mhz_median"] / 1e3
    except Exception:  # noqa: BLE001
        pass
    # durations: the last dispatch of every kernel in the plain trace
    dur = {}
    trace = find(src, "trace", "kernel_trace.csv")
    with open(trace) as f:
        for r in csv.DictReader(f):
            dur[short(r["Kernel_Name"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    factors = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        cal = collections.defaultdict(list)
        with open(find(src, "cal_" + c, "counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                cal[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
        copy = [v for k, v in cal.items() if "copyBuffer" in k][0]
        if c == "FETCH_SIZE":
            factors[c] = GIB / (sum(copy) / len(copy))
        else:
            zero = [v for k, v in cal.items() if "elementwise" in k][0]
            factors[c] = GIB / ((sum(copy) / len(copy) + sum(zero) / len(zero)) / 2)
    counters = collections.defaultdict(dict)
    for sub in ["FETCH_SIZE", "WRITE_SIZE"] + sorted(os.path.basename(p) for p in glob.glob(os.path.join(src, "sq*")) if os.path.isdir(p)):
        path = find(src, sub, "counter_collection.csv")
        if not path:
            continue
        for k, cs in last_values(path).items():
            counters[k].update(cs)
    out = {"source": src, "reads": n_reads, "clock_ghz": clock_ghz, "calibration": factors,
           "collected_at_head": open(os.path.join(src, "head.txt")).read().strip() if os.path.exists(os.path.join(src, "head.txt")) else None,
           "how": "rocprofv3 --kernel-trace --pmc <set>, one set per pass; the LAST dispatch of every kernel (the second of two passes of "
                  "tools/stage_counters_run.py); FETCH_SIZE / WRITE_SIZE in KiB x the factors the 1 GiB zero_ / copy_ kernels give in the "
                  "same collection", "kernels": {}}
    for k in sorted(counters):
        if not k.startswith("pg_"):
            continue
        c = counters[k]
        t = dur.get(k)
        fetch = c.get("FETCH_SIZE", 0.0) * 1024.0 * factors["FETCH_SIZE"]
        write = c.get("WRITE_SIZE", 0.0) * 1024.0 * factors["WRITE_SIZE"]
        row = {"duration_us": None if t is None else t * 1e6, "hbm_fetch_bytes": fetch, "hbm_write_bytes": write,
               "counters": {x: c[x] for x in sorted(c) if x not in ("FETCH_SIZE", "WRITE_SIZE")}}
        if t:
            row["hbm_gbs"] = (fetch + write) / t / 1e9
            row["hbm_frac_of_peak"] = (fetch + write) / t / HBM_PEAK
            valu = c.get("SQ_INSTS_VALU")
            if valu is not None:
                row["valu_issue_frac"] = valu * 4.0 / (SIMDS * clock_ghz * 1e9 * t)
            waves, wcyc = c.get("SQ_WAVES"), c.get("SQ_WAVE_CYCLES")
            if wcyc is not None:
                # SQ_WAVE_CYCLES counts quad-cycles x resident wavefronts, summed over the shader engines' SQs
                row["mean_resident_waves_per_simd"] = wcyc * 4.0 / (SIMDS * clock_ghz * 1e9 * t)
            if waves:
                row["waves"] = waves
            lds = c.get("SQ_INSTS_LDS")
            if lds is not None and valu:
                row["lds_per_valu"] = lds / valu
            b = max((row.get("hbm_frac_of_peak", 0.0), "hbm"), (row.get("valu_issue_frac", 0.0), "valu issue"))
            row["bound"] = ("%s (%.0f %% of its peak)" % (b[1], 100 * b[0])) if b[0] >= 0.3 else \
                "neither HBM nor VALU issue (%.0f %% / %.0f %% of peak): latency of dependent loads / LDS round trips at the resident wavefront count" % (
                    100 * row.get("hbm_frac_of_peak", 0.0), 100 * row.get("valu_issue_frac", 0.0))
            row["reads_per_s_if_every_read_ran_it"] = n_reads / t
        out["kernels"][k] = row
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, r in out["kernels"].items():
        print("%-34s %9.1f us  hbm %5.1f %%  valu %5.1f %%  %s" % (k, r.get("duration_us") or 0, 100 * r.get("hbm_frac_of_peak", 0), 100 * r.get("valu_issue_frac", 0), r.get("bound", "")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 200000)
