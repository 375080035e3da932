#!/usr/bin/env python
"""bench.py -- headline benchmark of the read -> variant-graph realignment path on MI355X.

A "step" is one pass of the hot path (4 graph fills + strand pick + traceback per read, i.e. the default
grmpy cascade GraphAligner::alignRead(AF_ALL)) over one batch of synthetic reads that is already
resident in HBM when the timed region starts, followed by the count path (read filters, node/edge/sequence
support, per-fragment union, per-site counters).

Workload at N=1 = BASELINE.json configs[1]: 1 DEL graph (200 bp flanks, 100 bp deletion; nodes
201/100/201 bp, G = 502), 1 000 000 synthetic 150 bp reads.  With N GPUs
every rank runs the same-sized batch on its own GPU (weak scaling, reads/sites are independent; no
data-path collective; the only collective is the RCCL reduce of a small per-rank tally table at the end
of each step).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel (pg_fill_kernel): algorithmic bytes of SURVEY.md 8(d)
                  (B_alg = 6*L*G + L + 64 per read) / HIP-event duration of its launches / 8 TB/s
  cpu_baseline -- the reference's own gssw.c (oracle/_ref, kind "reference") or the plain-C port, timed
                  on this host's cores on a bounded sample of the same reads, chunk-per-thread as the
                  reference parallelises (Align.cpp:114-156)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU per step (config 2: 1M)")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3"],
                    help="config2 = BASELINE configs[1] (headline); config3 = configs[2]: mixed DEL/INS sites, 30x")
    ap.add_argument("--sites", type=int, default=2000, help="sites per GPU for --workload config3")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--workspace-gib", type=float, default=64.0,
                    help="HBM budget for traceback state (two halves: trace of chunk i overlaps fill of chunk i+1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-batches", type=int, default=4,
                    help="batches of the PCIe-inclusive streaming leg (0 = skip; reported beside the headline value)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_r01.json"),
                    help="PMC-derived HBM bytes per fill launch (written by tools/pmc_traffic.py), optional")
    return ap.parse_args()


def cpu_baseline(site, arr, seconds):
    """Times the CPU checker on a bounded sample of the same reads (rank 0, N=1 only).

    The thread count is chosen by a short probe (gssw allocates and zeroes 11 buffers per node per fill,
    gssw.c:186-212, so it stops scaling long before a big host runs out of cores); `cores` reports the
    thread count actually used for the timed sample."""
    from oracle import oracle as orc
    if orc.have_ref():
        chk, kind, label = orc.RefOracle(), "reference", "reference gssw.c (oracle/_ref)"
    else:
        chk, kind, label = orc.PortOracle(), "port", "plain-C restatement (oracle/pg_oracle.c)"
    ncpu = os.cpu_count() or 1
    reads = [row.tobytes().decode() for row in arr[:min(len(arr), 400000)]]
    best_t, best_rate, worse = 1, 0.0, 0
    t = 1
    chk.align_batch(site.seqs, site.edges, reads[:64], threads=1, want_cigars=True)  # warm-up
    while t <= ncpu and worse < 2:
        n = min(len(reads), 256 * t)
        t0 = time.perf_counter()
        chk.align_batch(site.seqs, site.edges, reads[:n], threads=t, want_cigars=True)
        rate = n / max(time.perf_counter() - t0, 1e-6)
        log("cpu probe: %d threads -> %.0f reads/s" % (t, rate))
        if rate > best_rate:
            best_t, best_rate, worse = t, rate, 0
        else:
            worse += 1
        t *= 2
    done, spent, pos = 0, 0.0, 0
    # short slices: gssw's per-fill malloc/free churn degrades long single calls on glibc (heap growth per thread)
    slice_n = 512 * best_t
    while spent < seconds and pos < len(reads):
        n = min(slice_n, len(reads) - pos)
        t0 = time.perf_counter()
        chk.align_batch(site.seqs, site.edges, reads[pos:pos + n], threads=best_t, want_cigars=True)
        spent += time.perf_counter() - t0
        done += n
        pos += n
    return {"value": done / spent, "unit": "reads/s", "cores": best_t, "kind": kind,
            "sample": "%d of the same config-2 reads, %s, %d threads of %d host CPUs (one aligner per contiguous "
                      "chunk, Align.cpp:114-156; thread count picked by a throughput probe), %.1f s"
                      % (done, label, best_t, ncpu, spent)}


def log(msg):
    if os.environ.get("PG_BENCH_VERBOSE"):
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        try:
            import torch
        except Exception:  # torch is only plumbing here (barrier + device sync)
            torch = None

    from paragraph_amd import capi, synth

    # ---- workload: config 2, a different read seed per rank (weak scaling) -------------------------
    log("generating reads")
    L = args.read_len
    ctx = capi.Context(local_rank, workspace_bytes=int(args.workspace_gib * (1 << 30)))
    if args.workload == "config2":
        site, arr = synth.config2_reads_packed(args.reads, read_len=args.read_len, seed=2 + rank)
        log("reads generated")
        G = site.total_len
        n_sites = 1
        b_alg_total = args.reads * (6 * L * G + L + 64)
        graphs = ctx.upload_graphs([(site.seqs, site.edges)])
        graphs.set_labels([site.labels])
        batch = ctx.new_batch()
        t0 = time.perf_counter()
        batch.upload(graphs, synth.packed_to_capi(arr))
        # mates: reads 2k and 2k+1 form fragment k (counts are per fragment, ReadCounting.cpp:52-94)
        batch.set_fragments(np.arange(args.reads, dtype=np.uint32) // 2)
    else:
        sites = synth.mixed_sites(args.sites, seed=3 + rank, read_len=L)
        log("sites generated")
        site = sites[0].site
        arr = np.concatenate([s.reads for s in sites])
        args.reads = len(arr)
        n_sites = len(sites)
        G = float(np.mean([s.site.total_len for s in sites]))
        b_alg_total = int(sum(len(s.reads) * (6 * L * s.site.total_len + L + 64) for s in sites))
        graphs = ctx.upload_graphs([(s.site.seqs, s.site.edges) for s in sites])
        graphs.set_labels([s.site.labels for s in sites])
        batch = ctx.new_batch()
        t0 = time.perf_counter()
        gor = np.concatenate([np.full(len(s.reads), i, dtype=np.uint32) for i, s in enumerate(sites)])
        batch.upload(graphs, synth.packed_to_capi(arr), gor)
        batch.set_fragments(np.concatenate([s.fragment for s in sites]), np.concatenate([s.is_reverse for s in sites]))
    ctx.sync()
    t_upload = time.perf_counter() - t0
    log("uploaded in %.2fs" % t_upload)

    # per-site counter table {count, READS, FWD, REV} x (nodes, edges, sequence sets) + filter tallies: a torch
    # tensor so that the final reduce is ONE RCCL all-reduce over xGMI on device memory, no host hop
    n_counters = int(graphs.layout.n_counters)
    counts_t = None
    if torch is not None and torch.cuda.is_available():
        counts_t = torch.zeros(n_counters, dtype=torch.int32, device="cuda:%d" % local_rank)

    def barrier():
        ctx.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def step():
        batch.align(capi.AF_ALL)
        if counts_t is not None:
            counts_t.zero_()
            torch.cuda.current_stream().synchronize()
            batch.count(remove_nonuniq=True, bad_align_frac=0.8, d_counts=counts_t.data_ptr())
        else:
            batch.count(remove_nonuniq=True, bad_align_frac=0.8)
        if world > 1:
            # the only collective of the path: sum of the per-site counters (RCCL over xGMI)
            ctx.sync()
            dist.all_reduce(counts_t)

    for _ in range(args.warmup):
        step()
    barrier()
    log("warmup done")
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    log("timed region %.3fs" % elapsed)
    tim = ctx.timing()
    ctx.timing_enable(False)

    # PCIe-inclusive leg (not the headline value): download of results + ops
    t0 = time.perf_counter()
    res, ops = batch.download()
    t_download = time.perf_counter() - t0
    site_counts = None
    if counts_t is not None:
        tab = counts_t.cpu().numpy().view(np.uint32)
        site_counts = capi.decode_counts(graphs, tab)[0]
    log("downloaded in %.2fs" % t_download)

    # PCIe-inclusive leg, streaming form: host buffers -> device -> results on the host, two batch objects; the
    # upload of batch i+1 and the download of batch i-1 run on the copy stream under the kernels of batch i
    t_stream = None
    if world == 1 and args.workload == "config2" and args.stream_batches > 0:
        packed = synth.packed_to_capi(arr)
        frag = np.arange(args.reads, dtype=np.uint32) // 2
        bb = [ctx.new_batch(), ctx.new_batch()]
        for b in bb:  # allocate once (steady state)
            b.upload(graphs, packed)
            b.set_fragments(frag)
        ctx.sync()
        t0 = time.perf_counter()
        pending = None
        for i in range(args.stream_batches):
            b = bb[i & 1]
            b.upload(graphs, packed)
            b.set_fragments(frag)
            b.align(capi.AF_ALL)
            b.count(remove_nonuniq=True, bad_align_frac=0.8)
            if pending is not None:
                pending.download()
                pending.download_counts()
            pending = b
        pending.download()
        pending.download_counts()
        ctx.sync()
        t_stream = (time.perf_counter() - t0) / args.stream_batches
        log("streaming leg: %.3fs per batch" % t_stream)
        for b in bb:
            b.close()

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        reads_total = args.reads * world * args.steps
        value = reads_total / elapsed
        b_alg = b_alg_total / args.reads  # 6*L*G + L + 64 per read
        fill_s = tim["fill_ms"] / 1e3
        reads_per_fill_leg = args.reads * args.steps  # this rank's fill launches
        achieved_gbs = reads_per_fill_leg * b_alg / fill_s / 1e9 if fill_s > 0 else 0.0
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                with open(args.traffic_json) as f:
                    per_read = json.load(f).get("hbm_bytes_per_read")
                # PMC bytes (rocprofv3 FETCH_SIZE x measured 2.0 + WRITE_SIZE, separate passes, profiles/pmc_r01)
                # per read x the reads one launch of THIS run processes
                traffic = per_read * reads_per_fill_leg / max(1, tim["fill_launches"])
            except Exception:
                traffic = None
        out = {
            "metric": "150bp reads aligned/sec (whole node)",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16x2 packed (8-bit scores)",
            "data": "synthetic",
            "config": {
                "workload": ("configs[1]: 1 DEL graph (200bp flanks, nodes 201/100/201), %d synthetic %dbp reads per GPU, "
                             "GraphAligner::alignRead(AF_ALL) = 4 fills + strand pick + traceback per read, then "
                             "filters + node/edge/sequence counts" % (args.reads, L)) if args.workload == "config2" else
                            ("configs[2]: %d mixed DEL/longDEL/INS sites per GPU, 30x paired %dbp reads (%d reads), align + "
                             "count" % (n_sites, L, args.reads)),
                "sites_per_gpu": n_sites, "sites_per_s": n_sites * world * args.steps / elapsed,
                "reads_per_gpu": args.reads, "read_len": L, "graph_len": G, "parallelism": "reads x%d" % world,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "pg_fill_kernel<%d, false>" % (2 * ((L + 31) // 32)),
                "launches": int(tim["fill_launches"]),
                "avg_launch_ms": tim["fill_ms"] / max(1, tim["fill_launches"]),
                "alg_bytes_per_read": b_alg,
                "alg_bytes_per_launch": b_alg * reads_per_fill_leg / max(1, tim["fill_launches"]),
                "gcups": tim["cells"] / fill_s / 1e9 if fill_s > 0 else 0.0,
                "trace_bytes_written_per_read": tim["trace_bytes"] / max(1, reads_per_fill_leg),
            },
            "kernel_ms": {"fill": tim["fill_ms"], "trace": tim["trace_ms"]},
            "pcie_inclusive": {
                "upload_s": t_upload, "download_s": t_download,
                "reads_per_s": args.reads / (t_upload + elapsed / args.steps + t_download),
                "streaming_s_per_batch": t_stream,
                "streaming_reads_per_s": (args.reads / t_stream) if t_stream else None,
                "note": "upload_s is the FIRST upload (allocations included); streaming = double-buffered steady state, "
                        "host arrays in, results + ops + supports + counts out",
            },
        }
        if site_counts is not None and args.workload == "config2":
            out["counts"] = {"edges": {"%s_%s" % (site.names[a], site.names[b]): c[0]
                                       for (a, b), c in site_counts["edge_counts"].items()},
                             "sequences": {k: v[0] for k, v in site_counts["seq_counts"].items()},
                             "tallies": site_counts["tallies"],
                             "note": "fragment counts of the last step, summed over %d rank(s)" % world}
        if world == 1 and not args.no_cpu_baseline and args.workload == "config2":
            out["cpu_baseline"] = cpu_baseline(site, arr, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
