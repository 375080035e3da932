"""One mode of the workflow under a kernel trace: python tools/e2e/mode_trace.py run <dir> <passes> [key=value ...] runs the passes
(after a warm-up pass) on a dataset made in <dir>; `summary <trace.csv> <passes + 1> <seconds per pass>` prints what the device did
per pass: per-kernel time, launches, and how much of the wall clock had at least one kernel running."""
import collections
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    from paragraph_amd import synth_e2e, workflow
    d, passes = sys.argv[2], int(sys.argv[3])
    opts = {"threads": 16}
    for kv in sys.argv[4:]:
        k, v = kv.split("=")
        opts[k] = (v not in ("0", "false")) if k.endswith("matching") else int(v)
    os.makedirs(d, exist_ok=True)
    if not os.path.exists(d + "/graphs.txt"):
        synth_e2e.make_dataset(d, n_sites=10000, procs=os.cpu_count() or 1)
    graphs = [l.strip() for l in open(d + "/graphs.txt") if l.strip()]
    workflow.genotype_graphs_to_file(d + "/ref.fa", d + "/manifest.txt", graphs, d + "/out.json", **opts)
    t0 = time.time()
    for _ in range(passes):
        workflow.genotype_graphs_to_file(d + "/ref.fa", d + "/manifest.txt", graphs, d + "/out.json", **opts)
    s = (time.time() - t0) / passes
    print(json.dumps({"options": opts, "sites": len(graphs), "passes": passes, "s_per_pass": s, "sites_per_s": len(graphs) / s}))


def summary():
    f, passes, s_per_pass = sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    tot = collections.defaultdict(float)
    cnt = collections.Counter()
    spans = []
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        tot[n] += (e - b) / 1e6
        cnt[n] += 1
        spans.append((b, e))
    spans.sort()
    busy, cur_b, cur_e = 0, None, None
    for b, e in spans:
        if cur_e is None or b > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_b
            cur_b, cur_e = b, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_b
    top = sorted(tot, key=lambda k: -tot[k])[:10]
    # the seed stream's kernels beside the fills: duration by how much of the kernel a fill was running for
    fills = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if "pg_fill_kernel" in r["Kernel_Name"])
    beside = {}
    for want in ("pg_revcomp_kernel", "pg_path_kernel", "pg_fragment_kernel", "pg_support_kernel"):
        rows = []
        for r in csv.DictReader(open(f)):
            if want not in r["Kernel_Name"]:
                continue
            b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            ov = sum(max(0, min(e, fe) - max(b, fb)) for fb, fe in fills if fe > b and fb < e)
            rows.append(((e - b) / 1e3, min(1.0, ov / max(1, e - b)), int(r.get("Grid_Size_X", 0) or 0)))
        if rows:
            alone = sorted(d for d, o, g in rows if o < 0.05)
            under = sorted(d for d, o, g in rows if o > 0.95)
            med = lambda v: round(v[len(v) // 2], 1) if v else None
            beside[want] = {"launches": len(rows), "no_fill_running": {"n": len(alone), "median_us": med(alone)},
                            "a_fill_running_throughout": {"n": len(under), "median_us": med(under)},
                            "median_grid_threads": sorted(g for d, o, g in rows)[len(rows) // 2]}
    # the fill launches: wavefronts (= workgroups) per launch against the 4 096 the device holds at once (256 CUs x 4 SIMDs x 4)
    fl = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r.get("Grid_Size_X", 0) or 0) // 64)
          for r in csv.DictReader(open(f)) if "pg_fill_kernel" in r["Kernel_Name"]]
    fill_launches = None
    if fl:
        w = sorted(g for d, g in fl)
        d = sorted(d for d, g in fl)
        q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
        fill_launches = {"launches": len(fl), "wavefronts": {"p10": q(w, 0.1), "median": q(w, 0.5), "p90": q(w, 0.9), "sum_per_pass": sum(w) / passes},
                         "us": {"p10": round(q(d, 0.1), 1), "median": round(q(d, 0.5), 1), "p90": round(q(d, 0.9), 1)},
                         "wavefront_slots": 4096}
    # between two fills: how long, and what ran meanwhile (the fill stream is in order: the next fill was either not queued yet, or
    # waiting for its batch's upload / for a workspace region a traceback still reads)
    allk = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    fk = [k for k in allk if "pg_fill_kernel" in k[2]]
    gaps = []
    for (b0, e0, _), (b1, e1, _) in zip(fk, fk[1:]):
        if b1 > e0 and b1 - e0 < 20e6:  # (not the pause between passes)
            inside = collections.Counter()
            for b, e, n in allk:
                if e > e0 and b < b1 and "pg_fill_kernel" not in n:
                    inside[n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]] += min(e, b1) - max(b, e0)
            gaps.append((b1 - e0, inside))
    gap_summary = None
    if gaps:
        g = sorted(x for x, _ in gaps)
        tot_busy = collections.Counter()
        for _, bz in gaps:
            tot_busy.update(bz)
        gap_summary = {"gaps": len(g), "sum_ms_per_pass": sum(g) / 1e6 / passes, "median_us": g[len(g) // 2] / 1e3, "p90_us": g[int(0.9 * len(g))] / 1e3,
                       "overlapping_fills": sum(1 for (b0, e0, _), (b1, e1, _) in zip(fk, fk[1:]) if b1 <= e0),
                       "kernel_ms_inside_gaps_per_pass": {k: round(v / 1e6 / passes, 2) for k, v in tot_busy.most_common(6)}}
    print(json.dumps({"fill_launches": fill_launches, "between_fills": gap_summary, "ms_per_pass_wall": s_per_pass * 1e3, "ms_per_pass_with_a_kernel_running": busy / 1e6 / passes,
                      "kernel_ms_per_pass": {k: round(tot[k] / passes, 2) for k in top},
                      "launches_per_pass": {k: round(cnt[k] / passes, 1) for k in top},
                      "avg_us": {k: round(tot[k] / cnt[k] * 1e3, 1) for k in top}, "beside_the_fills": beside}, indent=1))


if __name__ == "__main__":
    {"run": run, "summary": summary}[sys.argv[1]]()
