"""Throughput of the optional cascade stages (path / k-mer / klib) and of the gssw stage on config-2 reads, each alone, and of the
composed cascades with the filter chain between the stages and the hand-over decided on the device (pg_batch_retire_mapped):
`paragraph`'s default path + gssw (src/c++/main/paragraph.cpp:60-61) and all four stages.  `predicted_reads_per_s` = what the
stage rates give for the same split of the reads (every read through the first stage, the rest through the next, ...).
Usage: python tools/stage_probe.py [n_reads]   (prints one JSON object)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from paragraph_amd import capi, synth  # noqa: E402


def timed(ctx, fn, reps=3):
    fn()
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    ctx = capi.Context(0, workspace_bytes=64 << 30)
    site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
    graphs = ctx.upload_graphs([(site.seqs, site.edges)])
    paths = [[[0, 1, 2], [0, 2]]]
    graphs.build_path_index(32)
    graphs.build_kmer_index(paths, 16)
    graphs.build_klib_index(paths)
    b = ctx.new_batch()
    b.upload(graphs, synth.packed_to_capi(arr))
    out = {"reads": n, "read_len": 150, "graph_len": int(site.total_len)}
    b.align(capi.AF_ALL)  # (the device's clocks are up before the first stage is timed: the path stage is 4 ms of work)
    ctx.sync()
    for name, fn in (("path", b.path_align), ("kmer", b.kmer_align), ("klib", b.klib_align), ("gssw", lambda: b.align(capi.AF_ALL))):
        s = timed(ctx, fn, reps=8 if name == "path" else 3)
        flags = None
        if name != "gssw":
            flags = fn()
        out[name] = {"s_per_batch": s, "reads_per_s": n / s,
                     "mapped_frac": None if flags is None else float(np.mean((flags & 1) != 0))}
    assert graphs.klib_error() == 0
    # ---- composed cascades: stage, count pass (NonUniq + BadAlign), hand-over on the device, next stage on what is left.
    # Two batch objects holding the same reads take turns, as in bench.py: the count pass and the hand-over of one run under the
    # stages of the other -- the steady state of a workflow whose batches are different reads.
    graphs.set_labels([site.labels])
    b2 = ctx.new_batch()
    b2.upload(graphs, synth.packed_to_capi(arr))
    for x in (b, b2):
        x.set_fragments(np.arange(n, dtype=np.uint32) // 2)
    keep_flags = capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS

    def cascade(stages):
        """(start, finish): the seed stages + filter chain + device hand-over of a batch are queued by start(), its gssw stage and
        final count pass by finish().  The loop below queues start(next batch) BEFORE finish(this batch): by the time the host
        asks for this batch's per-graph counts (the one wait of the hand-over) they have long arrived, and the device's main
        stream goes path, fills, path, fills ... without a gap -- what lanes of a workflow do for each other."""
        def start(x):
            x.set_active(None)
            keep = 0
            for name in stages:
                if name == "path":
                    x.path_align(fetch_flags=False)
                elif name == "kmer":
                    x.kmer_align(keep, fetch_flags=False)
                else:
                    x.klib_align(keep, fetch_flags=False)
                x.count(remove_nonuniq=True, bad_align_frac=0.8)
                x.retire_mapped()
                keep = capi.AF_KEEP_RESULTS

        def finish(x):
            x.align(keep_flags)
            x.count(remove_nonuniq=True, bad_align_frac=0.8)
        return start, finish

    def pipelined(start, finish, reps):
        start(b)
        ctx.sync()
        t = time.perf_counter()
        cur, nxt = b, b2
        for _ in range(reps):  # steady state: one batch started and not finished at either end of the timed region
            start(nxt)
            finish(cur)
            cur, nxt = nxt, cur
        ctx.sync()
        s = (time.perf_counter() - t) / reps
        finish(cur)
        ctx.sync()
        return s, cur

    for key, stages in (("cascade_path_gssw", ["path"]), ("cascade_all_four", ["path", "kmer", "klib"])):
        start, finish = cascade(stages)
        pipelined(start, finish, 2)  # warm-up
        s, x = pipelined(start, finish, 7)
        res, _, _, sup, _ = x.download_all(want_table=False)
        by = {"path": int(((res["status"] & capi.STATUS_PATH_ALIGNER) != 0).sum()), "kmer": int(((res["status"] & capi.STATUS_KMER_ALIGNER) != 0).sum()),
              "klib": int(((res["status"] & capi.STATUS_KLIB_ALIGNER) != 0).sum())}
        left, t_pred = n, 0.0
        for name in stages:
            t_pred += left / out[name]["reads_per_s"]
            left -= by[name]
        t_pred += left / out["gssw"]["reads_per_s"]
        out[key] = {"s_per_batch": s, "reads_per_s": n / s, "predicted_reads_per_s": n / t_pred, "vs_predicted": t_pred / s,
                    "reads_by_stage": dict(by, gssw=left), "mapped": int((sup["status"] == 1).sum()),
                    "note": "two batch objects, the next one's seed stages queued before this one's gssw stage; predicted = every read "
                            "through the first stage at its rate, the rest through the next, ..."}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
