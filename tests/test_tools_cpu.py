"""The measuring tools that turn rocprofv3 / workflow traces into the numbers DESIGN.md quotes: run on synthetic traces (no GPU)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X"])
        for r in rows:
            w.writerow(r)


def test_mode_trace_summary_counts_gaps_and_neighbours(tmp_path):
    # two fills of 1 ms with 0.5 ms between them, in which a traceback runs for 0.2 ms; a path kernel beside the first fill and
    # one with the device to itself
    us = 1000
    rows = [("void pg_fill_kernel<10, false, 16>(PgFillArgs)", 0, 1000 * us, 64 * 5000),
            ("void pg_trace_kernel<10, false, 16>(PgTraceArgs)", 1100 * us, 1300 * us, 64 * 100),
            ("void pg_fill_kernel<10, false, 16>(PgFillArgs)", 1500 * us, 2500 * us, 64 * 5000),
            ("(anonymous namespace)::pg_path_kernel((anonymous namespace)::PathArgs)", 100 * us, 900 * us, 64 * 50),
            ("(anonymous namespace)::pg_path_kernel((anonymous namespace)::PathArgs)", 3000 * us, 3300 * us, 64 * 50)]
    t = tmp_path / "t_kernel_trace.csv"
    _trace(t, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e", "mode_trace.py"), "summary", str(t), "1", "0.004"],
                         capture_output=True, text=True, check=True).stdout
    d = json.loads(out)
    assert d["fill_launches"]["launches"] == 2 and d["fill_launches"]["wavefronts"]["median"] == 5000
    g = d["between_fills"]
    assert g["gaps"] == 1 and abs(g["sum_ms_per_pass"] - 0.5) < 1e-9 and g["overlapping_fills"] == 0
    assert abs(g["kernel_ms_inside_gaps_per_pass"]["pg_trace_kernel"] - 0.2) < 1e-9
    p = d["beside_the_fills"]["pg_path_kernel"]
    assert p["launches"] == 2
    assert p["a_fill_running_throughout"] == {"n": 1, "median_us": 800.0} and p["no_fill_running"] == {"n": 1, "median_us": 300.0}
    assert abs(d["ms_per_pass_with_a_kernel_running"] - (1.0 + 0.2 + 1.0 + 0.3)) < 1e-9
    assert d["kernel_ms_per_pass"]["pg_fill_kernel<10, false, 16>"] == 2.0
