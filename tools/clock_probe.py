#!/usr/bin/env python
"""Samples the engine clock the GPU reports while a command runs (for the sustained clock under the fill's load).

usage: tools/clock_probe.py out.json -- <command ...>

Three sources, whichever answer on the box: the sysfs DPM table (pp_dpm_sclk: the level marked '*'), `rocm-smi
--showclocks --json`, `amd-smi metric --clock --json`.  The command's stdout is passed through; the samples (time since
start, MHz) and their summary go to out.json.  The probe never touches HIP itself."""
import glob
import json
import re
import subprocess
import sys
import threading
import time


def sysfs_sclk():
    out = []
    for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            with open(path) as f:
                for line in f:
                    m = re.match(r"\s*\d+:\s*(\d+)\s*[Mm][Hh]z\s*\*", line)
                    if m:
                        out.append(int(m.group(1)))
        except OSError:
            pass
    return out


def smi_json(cmd):
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10)
        return json.loads(p.stdout.decode())
    except Exception:  # noqa: BLE001
        return None


def find_mhz(doc, keys):
    """every number found under a key whose name matches one of `keys` (nested dicts / lists), as MHz"""
    found = []

    def walk(x, under):
        if isinstance(x, dict):
            for k, v in x.items():
                walk(v, under or any(s in str(k).lower() for s in keys))
        elif isinstance(x, list):
            for v in x:
                walk(v, under)
        elif under:
            m = re.match(r"\(?\s*(\d+(?:\.\d+)?)\s*(?:[Mm][Hh]z)?\)?$", str(x).strip())
            if m:
                found.append(float(m.group(1)))

    walk(doc, False)
    return found


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    samples = {"sysfs_pp_dpm_sclk": [], "rocm_smi_sclk": [], "amd_smi_gfx": []}
    raw_first = {}
    stop = threading.Event()
    t0 = time.perf_counter()

    def fast():
        while not stop.is_set():
            v = sysfs_sclk()
            if v:
                samples["sysfs_pp_dpm_sclk"].append([round(time.perf_counter() - t0, 3)] + v)
            time.sleep(0.05)

    def slow():
        while not stop.is_set():
            d = smi_json(["rocm-smi", "--showclocks", "--json"])
            if d is not None:
                raw_first.setdefault("rocm_smi", d)
                v = find_mhz(d, ["sclk"])
                if v:
                    samples["rocm_smi_sclk"].append([round(time.perf_counter() - t0, 3)] + v)
            d = smi_json(["amd-smi", "metric", "--clock", "--json"])
            if d is not None:
                raw_first.setdefault("amd_smi", d)
                v = find_mhz(d, ["gfx"])
                if v:
                    samples["amd_smi_gfx"].append([round(time.perf_counter() - t0, 3)] + v[:32])

    th = [threading.Thread(target=fast, daemon=True), threading.Thread(target=slow, daemon=True)]
    for t in th:
        t.start()
    rc = subprocess.call(cmd)
    wall = time.perf_counter() - t0
    stop.set()
    for t in th:
        t.join(timeout=15)
    summary = {}
    for k, rows in samples.items():
        vals = [x for r in rows for x in r[1:]]
        if vals:
            vals.sort()
            summary[k] = {"samples": len(rows), "min_mhz": vals[0], "median_mhz": vals[len(vals) // 2], "max_mhz": vals[-1]}
    with open(out_path, "w") as f:
        json.dump({"command": cmd, "rc": rc, "wall_s": wall, "summary": summary, "samples": samples, "first_raw_answers": raw_first}, f, indent=1)
    print(json.dumps(summary), file=sys.stderr)
    return rc


if __name__ == "__main__":
    sys.exit(main())
