#!/usr/bin/env python
"""bench.py -- headline benchmark of the read -> variant-graph realignment path on MI355X.

A "step" is one pass of the hot path (4 graph fills + strand pick + traceback per read, i.e. the default
grmpy cascade GraphAligner::alignRead(AF_ALL)) over one batch of synthetic reads that is already
resident in HBM when the timed region starts, followed by the count path (read filters, node/edge/sequence
support, per-fragment union, per-site counters) and -- with N > 1 ranks -- the path's only collective, ONE
all-reduce of the per-site counter table (RCCL over xGMI).

Workloads
  config2 (default, headline) = BASELINE.json configs[1]: 1 DEL graph (200 bp flanks, 100 bp deletion; nodes
      201/100/201 bp, G = 502), 1 000 000 synthetic 150 bp reads per GPU.  With N GPUs every rank aligns its own
      1 M reads ("scaling": "weak").
  config3 = BASELINE.json configs[2] / configs[3]: ONE set of 10 000 mixed DEL / long-DEL / INS sites with 30x paired
      reads; with N GPUs the SAME set is partitioned over the ranks by paragraph_amd.dist.partition_sites (weights =
      sum of read length x graph length), every rank aligns + counts its shard into the global-layout table, one
      all-reduce ("scaling": "strong").  Rank 0 checks the reduced table against a 1-rank pass over all sites.
  The default run times config2 and then runs a short config3 leg, reported as "sites" in the same JSON line.

Launch: `python bench.py --gpus N` spawns its N ranks itself (torch.distributed.run, 127.0.0.1); under an external
`torch.distributed.run` (WORLD_SIZE in the environment) it is one of the ranks.  On a box with fewer GPUs than ranks
the ranks share devices and the reduce runs over gloo -- same code path, used by the tests.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- dominant kernel (pg_fill_kernel): algorithmic bytes of SURVEY.md 8(d)
                  (B_alg = 6*L*G + L + 64 per read) / HIP-event duration of its launches / 8 TB/s
  cpu_baseline -- the reference's own gssw.c (oracle/_ref, kind "reference") or the plain-C port, timed on this
                  host's cores in a separate process (threads in one process AND one process per chunk; the better
                  one is `value`), N=1 only
  verified     -- the GPU results of the last timed step compared field by field and CIGAR by CIGAR with the
                  reference alignments the cpu_baseline leg computed for the same reads (exit status 3 on a mismatch)
  sites        -- the config3 leg (sites/s, strong scaling, reduce_equals_single)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
CIGAR_STRIDE = 128


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU per step (config 2: 1M)")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3"],
                    help="config2 = BASELINE configs[1] (headline, weak scaling); config3 = configs[2]/[3]: one set of mixed "
                         "DEL/INS sites at 30x, sharded over the ranks (strong scaling)")
    ap.add_argument("--sites", type=int, default=10000, help="sites of the config3 set (the whole job, all ranks together)")
    ap.add_argument("--sites-steps", type=int, default=3, help="timed passes of the config3 leg of the default run (0 = skip)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--workspace-gib", type=float, default=64.0,
                    help="HBM budget for traceback state (two halves: trace of chunk i overlaps fill of chunk i+1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-batches", type=int, default=16,
                    help="batches of the PCIe-inclusive streaming leg (0 = skip; reported beside the headline value)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_r02.json"),
                    help="PMC-derived HBM bytes per fill launch (written by tools/pmc_traffic.py), optional")
    # internal: the CPU baseline runs in its own process (it forks workers; the GPU process must not)
    ap.add_argument("--cpu-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-reads-file", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-out", help=argparse.SUPPRESS)
    return ap.parse_args()


def log(msg):
    if os.environ.get("PG_BENCH_VERBOSE"):
        print("[bench %.1fs rank %s] %s" % (time.perf_counter() - _T0, os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


# ---------------------------------------------------------------------------------------------------
# CPU baseline leg (own process: never touches HIP, free to fork)
# ---------------------------------------------------------------------------------------------------
def _shared_array(shape, dtype):
    import mmap
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    mm = mmap.mmap(-1, max(1, n * dt.itemsize))
    return np.frombuffer(mm, dtype=dt, count=n).reshape(shape)


def _run_procs(chk, site, off, bases, res, cig, lo, hi, procs):
    """Reads [lo, hi) cut into `procs` contiguous chunks, one forked worker (its own aligner, its own heap) per chunk."""
    procs = max(1, min(procs, hi - lo))
    step = (hi - lo + procs - 1) // procs
    pids = []
    for w in range(procs):
        b, e = lo + w * step, min(hi, lo + (w + 1) * step)
        if b >= e:
            break
        pid = os.fork()
        if pid == 0:
            code = 1
            try:
                chk.align_into(site.seqs, site.edges, off[b:e + 1], bases, res[b:e], cig[b:e], threads=1)
                code = 0
            finally:
                os._exit(code)
        pids.append(pid)
    bad = 0
    for pid in pids:
        _, st = os.waitpid(pid, 0)
        bad += st != 0
    if bad:
        raise RuntimeError("%d CPU baseline workers failed" % bad)


def _cpu_quota():
    """CPUs the cgroup lets this process use at once (cpu.max of cgroup v2: "<quota> <period>" or "max"), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def cpu_leg_main(args):
    """Times the CPU checker on a bounded sample of the same reads, two ways:
       (a) chunk-per-thread in one process -- how the reference parallelises (Align.cpp:114-156; gssw allocates and zeroes
           11 buffers per node per fill, gssw.c:186-212, so threads meet in the allocator), thread count from a probe with
           2 000 reads per thread;
       (b) chunk-per-process (same chunks, one forked worker each: no shared allocator), on all cores and on half of them.
    The better configuration aligns the timed sample from read 0 on; its results go to --cpu-out for the verification."""
    from oracle import oracle as orc
    from paragraph_amd import synth
    site = synth.config2_site()
    arr = np.load(args.cpu_reads_file, mmap_mode="r")
    n_all, L = arr.shape
    bases = np.ascontiguousarray(arr).reshape(-1)
    off = (np.arange(n_all + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
    if orc.have_ref():
        chk, kind, label = orc.RefOracle(), "reference", "reference gssw.c (oracle/_ref)"
    else:
        chk, kind, label = orc.PortOracle(), "port", "plain-C restatement (oracle/pg_oracle.c)"
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cpu_quota()  # a container may see every CPU of the host and still be held to a few of them
    res = _shared_array((n_all,), orc.RESULT_NP)
    cig = _shared_array((n_all, CIGAR_STRIDE), np.uint8)
    chk.align_into(site.seqs, site.edges, off[:65], bases, res[:64], cig[:64], threads=1)  # warm-up
    per_worker = 2000
    probes = []
    # (a) threads of one process
    best_t, best_t_rate, worse, t = 1, 0.0, 0, 1
    while t <= ncpu and worse < 2:
        n = min(n_all, per_worker * t, 200000)
        t0 = time.perf_counter()
        chk.align_into(site.seqs, site.edges, off[:n + 1], bases, res[:n], cig[:n], threads=t)
        rate = n / max(time.perf_counter() - t0, 1e-9)
        probes.append({"mode": "threads", "workers": t, "reads": n, "reads_per_s": rate})
        if rate > best_t_rate:
            best_t, best_t_rate, worse = t, rate, 0
        else:
            worse += 1
        t *= 2
    # (b) one process per chunk
    best_p, best_p_rate = 1, 0.0
    counts = {ncpu, max(1, ncpu // 2)}
    if quota and quota < ncpu:
        counts |= {max(1, int(round(quota))), max(1, int(round(2 * quota)))}
    for p in sorted(counts, reverse=True):
        n = min(n_all, per_worker * p, 200000)
        t0 = time.perf_counter()
        _run_procs(chk, site, off, bases, res, cig, 0, n, p)
        rate = n / max(time.perf_counter() - t0, 1e-9)
        probes.append({"mode": "processes", "workers": p, "reads": n, "reads_per_s": rate})
        if rate > best_p_rate:
            best_p, best_p_rate = p, rate
    use_procs = best_p_rate >= best_t_rate
    rate0 = max(best_p_rate, best_t_rate)
    target = int(min(n_all, max(per_worker, rate0 * args.cpu_seconds)))
    done, spent = 0, 0.0
    if use_procs:
        t0 = time.perf_counter()
        _run_procs(chk, site, off, bases, res, cig, 0, target, best_p)
        spent = time.perf_counter() - t0
        done = target
    else:
        # short slices: gssw's per-fill malloc/free churn degrades long single calls on glibc (heap growth per thread)
        slice_n = 512 * best_t
        while done < target:
            n = min(slice_n, target - done)
            t0 = time.perf_counter()
            chk.align_into(site.seqs, site.edges, off[done:done + n + 1], bases, res[done:done + n], cig[done:done + n], threads=best_t)
            spent += time.perf_counter() - t0
            done += n
    np.savez(args.cpu_out, res=np.array(res[:done]), cig=np.array(cig[:done]))
    workers = best_p if use_procs else best_t
    out = {"value": done / spent, "unit": "reads/s", "cores": workers, "kind": kind,
           "mode": "one process per chunk" if use_procs else "threads of one process",
           "threads_one_process": {"threads": best_t, "reads_per_s": best_t_rate},
           "process_per_chunk": {"processes": best_p, "reads_per_s": best_p_rate},
           "probes": probes, "host_cpus": ncpu, "cpu_quota_cores": quota,
           "sample": "the first %d of the same config-2 reads, %s, %s on %d of %d host CPUs (one aligner per contiguous "
                     "chunk, Align.cpp:114-156; probes with %d reads per worker), %.1f s%s"
                     % (done, label, "one forked process per chunk" if use_procs else "threads of one process", workers, ncpu,
                        per_worker, spent, "; the cgroup holds the process to %.0f CPUs" % quota if quota and quota < ncpu else "")}
    print(json.dumps(out))


def run_cpu_leg(args, arr):
    """Runs cpu_leg_main in a fresh interpreter; returns (cpu_baseline dict, reference results, reference CIGAR slots)."""
    tmp = tempfile.mkdtemp(prefix="pgbench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    reads_file, out_file = os.path.join(tmp, "reads.npy"), os.path.join(tmp, "cpu.npz")
    try:
        np.save(reads_file, arr)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", "--cpu-reads-file", reads_file, "--cpu-out", out_file,
               "--cpu-seconds", str(args.cpu_seconds)]
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, check=True)
        base = json.loads(p.stdout.decode().strip().splitlines()[-1])
        z = np.load(out_file)
        return base, z["res"], z["cig"]
    finally:
        for f in (reads_file, out_file):
            if os.path.exists(f):
                os.unlink(f)
        os.rmdir(tmp)


def verify_against_reference(capi, res, ops, ref_res, ref_cig):
    """GPU pg_results + rendered CIGARs vs the reference alignments of the same reads (the first len(ref_res) of the batch).
    Degenerate reads (reference score 0: empty CIGAR at position 0, undefined downstream in the reference) must come back as
    score 0 / no ops / status 1; their strand flag is not compared."""
    n = len(ref_res)
    g = res[:n]
    zero = ref_res["score"] == 0
    bad = np.zeros(n, dtype=bool)
    bad |= g["graph_pos"] != ref_res["graph_pos"]
    bad |= g["score"].astype(np.int32) != ref_res["score"]
    bad |= g["mapq"].astype(np.int32) != ref_res["mapq"]
    bad |= (g["is_unique"] != 0) != (ref_res["unique"] != 0)
    bad |= ((g["returned_reverse"] != 0) != (ref_res["returned_reverse"] != 0)) & ~zero
    bad |= ((g["status"] & 0xFF) != np.where(zero, 1, 0))
    for k in range(4):
        bad |= ((g["multi_mask"] >> k) & 1).astype(np.int32) != ref_res["multi"][:, k]
    gc = capi.render_cigars(g, ops, ref_cig.shape[1])
    bad |= (gc != ref_cig).any(axis=1)
    first = int(np.nonzero(bad)[0][0]) if bad.any() else None
    out = {"reads": int(n), "mismatches": int(bad.sum()),
           "fields": "graph_pos, score, mapq, unique, returned_reverse, multi[4], CIGAR string"}
    if first is not None:
        out["first_mismatch"] = {"read": first, "gpu_cigar": bytes(gc[first]).split(b"\0")[0].decode(),
                                 "ref_cigar": bytes(ref_cig[first]).split(b"\0")[0].decode(),
                                 "gpu": [int(g[first][f]) for f in ("graph_pos", "score", "mapq", "is_unique", "returned_reverse", "multi_mask", "status")],
                                 "ref": [int(ref_res[first][f]) for f in ("graph_pos", "score", "mapq", "unique", "returned_reverse")]}
    return out


# ---------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------
# config3: one site set, sharded
# ---------------------------------------------------------------------------------------------------
class SiteSet:
    """The config3 data set + its partition.  Every rank holds the graphs of ALL sites (the counter table then has the
    global layout on every rank and the all-reduce needs no index translation) and the reads of its own shard."""

    def __init__(self, sites, read_len, world):
        from paragraph_amd import dist as pgdist
        self.sites = sites
        self.L = read_len
        self.n_reads_site = np.array([len(s.reads) for s in sites], dtype=np.int64)
        self.g_len = np.array([s.site.total_len for s in sites], dtype=np.int64)
        self.weights = self.n_reads_site * read_len * self.g_len  # DP cells per site
        self.parts = pgdist.partition_sites(self.weights, world)

    def shard_arrays(self, idx):
        ss = [self.sites[i] for i in idx]
        arr = np.concatenate([s.reads for s in ss]) if ss else np.zeros((0, self.L), np.uint8)
        gor = np.concatenate([np.full(len(self.sites[i].reads), i, dtype=np.uint32) for i in idx]) if ss else np.zeros(0, np.uint32)
        frag = np.concatenate([s.fragment for s in ss]) if ss else np.zeros(0, np.uint32)
        rev = np.concatenate([s.is_reverse for s in ss]) if ss else np.zeros(0, np.uint8)
        return arr, gor, frag.astype(np.uint32), rev.astype(np.uint8)

    def b_alg(self, idx):
        return int(sum(int(self.n_reads_site[i]) * (6 * self.L * int(self.g_len[i]) + self.L + 64) for i in idx))


def run_sites_leg(args, env, ctx, capi, synth, sset, steps, warmup, timed_events):
    """K timed passes over the rank's shard of the site set + ONE all-reduce of the table per pass.  Returns a dict
    (rank 0 adds the check against the 1-rank table)."""
    torch, dist = env["torch"], env["dist"]
    rank, world = env["rank"], env["world"]
    from paragraph_amd import dist as pgdist
    graphs = ctx.upload_graphs([(s.site.seqs, s.site.edges) for s in sset.sites])
    graphs.set_labels([s.site.labels for s in sset.sites])
    n_counters = int(graphs.layout.n_counters)
    mine = sset.parts[rank]
    arr, gor, frag, rev = sset.shard_arrays(mine)
    batch = ctx.new_batch()
    batch.upload(graphs, synth.packed_to_capi(arr), gor)
    batch.set_fragments(frag, rev)
    table = torch.zeros(n_counters, dtype=torch.int32, device=env["device"])
    ctx.sync()

    def one_pass(b, t):
        ctx.counts_zero(t.data_ptr(), n_counters)
        b.align(capi.AF_ALL)
        b.count(remove_nonuniq=True, bad_align_frac=0.8, d_counts=t.data_ptr())
        ctx.sync_compute()
        if world > 1:
            pgdist.allreduce_counts(t)  # the only collective of the path
            torch.cuda.synchronize()

    for _ in range(warmup):
        one_pass(batch, table)
    env["barrier"]()
    if timed_events:
        ctx.timing_enable(True)
        ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass(batch, table)
    env["barrier"]()
    elapsed = env["max_over_ranks"](time.perf_counter() - t0)
    tim = None
    if timed_events:
        tim = ctx.timing()
        ctx.timing_enable(False)
    n_sites, n_reads = len(sset.sites), int(sset.n_reads_site.sum())
    out = {"config": "configs[2]/[3]: %d mixed DEL / long-DEL / INS sites, 30x paired %dbp reads (%d reads), the same set "
                     "partitioned over %d rank(s) by dist.partition_sites (LPT on reads x graph length); align + count + "
                     "1 all-reduce of the %d-counter table per pass" % (n_sites, sset.L, n_reads, world, n_counters),
           "sites": n_sites, "reads": n_reads, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
           "sites_per_s": n_sites * steps / elapsed, "reads_per_s": n_reads * steps / elapsed, "scaling": "strong",
           "shard_reads": [int(sset.n_reads_site[p].sum()) for p in sset.parts],
           "shard_imbalance": float(max(sset.weights[p].sum() for p in sset.parts) * world / max(1, sset.weights.sum())),
           "counters": n_counters, "reduce_equals_single": None}
    got = table.cpu().numpy().view(np.uint32).copy()
    tall = got[int(graphs.layout.tally_base):].reshape(-1, 4)
    out["tallies"] = {"aligned": int((tall[:, 0] & 0x7FFFFFFF).sum()), "mapped": int(tall[:, 1].sum()),
                      "bad_align": int(tall[:, 2].sum()), "nonuniq": int(tall[:, 3].sum())}
    out["table_sum"] = int(got.astype(np.uint64).sum())
    if world > 1 and rank == 0:
        # 1-rank pass over ALL sites on this rank's device: the reduced table must equal it entry for entry
        arr1, gor1, frag1, rev1 = sset.shard_arrays(np.arange(n_sites))
        b1 = ctx.new_batch()
        b1.upload(graphs, synth.packed_to_capi(arr1), gor1)
        b1.set_fragments(frag1, rev1)
        t1 = torch.zeros(n_counters, dtype=torch.int32, device=env["device"])
        ctx.counts_zero(t1.data_ptr(), n_counters)
        b1.align(capi.AF_ALL)
        b1.count(remove_nonuniq=True, bad_align_frac=0.8, d_counts=t1.data_ptr())
        ctx.sync()
        single = t1.cpu().numpy().view(np.uint32)
        out["reduce_equals_single"] = bool(np.array_equal(single, got))
        out["single_table_sum"] = int(single.astype(np.uint64).sum())
        b1.close()
    batch.close()
    graphs.close()
    return out, tim, sset.b_alg(mine), len(arr)


# ---------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------
def main_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from paragraph_amd import synth
    L = args.read_len
    headline3 = args.workload == "config3"
    want_sites = headline3 or args.sites_steps > 0

    # ---- data first: the generators fork workers, which must happen before this process touches HIP ------------
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    site = arr = None
    if not headline3:
        log("generating config2 reads")
        site, arr = synth.config2_reads_packed(args.reads, read_len=L, seed=2 + rank)
    sset = None
    if want_sites:
        log("generating the config3 site set")
        sites = synth.mixed_sites_parallel(args.sites, seed=3, procs=max(1, min(16, ncpu // max(1, world))), read_len=L)
        sset = SiteSet(sites, L, world)
    log("data ready")

    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    shared = world > ndev  # fewer GPUs than ranks: ranks share devices, the reduce runs over gloo (tests on a 1-GPU box)
    dev_index = local_rank % ndev
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    backend = None
    if world > 1:
        backend = "gloo" if shared else "nccl"
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
        log("process group: backend %s, world %d, device %d%s" % (backend, dist.get_world_size(), dev_index, " (shared)" if shared else ""))

    from paragraph_amd import capi
    from paragraph_amd import dist as pgdist
    ws_gib = args.workspace_gib
    if shared:
        ws_gib = min(ws_gib, max(8.0, 160.0 / ((world + ndev - 1) // ndev)))
    ctx = capi.Context(dev_index, workspace_bytes=int(ws_gib * (1 << 30)))

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if shared else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    env = {"torch": torch, "dist": dist, "rank": rank, "world": world, "device": device, "barrier": barrier,
           "max_over_ranks": max_over_ranks}
    dist_info = {"world": world, "backend": backend, "shared_device": bool(shared), "devices_visible": ndev,
                 "launcher": os.environ.get("PG_BENCH_LAUNCHER", "external torch.distributed.run" if world > 1 else "single process")}

    out = None
    if headline3:
        sites_out, tim, b_alg_mine, reads_mine = run_sites_leg(args, env, ctx, capi, synth, sset, args.steps, args.warmup, True)
        if rank == 0:
            fill_s = tim["fill_ms"] / 1e3
            achieved = b_alg_mine * args.steps / fill_s / 1e9 if fill_s > 0 else 0.0
            out = {
                "metric": "150bp reads aligned/sec (whole node)", "value": sites_out["reads_per_s"], "unit": "reads/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sites_out["ms_per_step"],
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u16x2 packed (8-bit scores)",
                "data": "synthetic",
                "config": {"workload": sites_out["config"], "sites": sites_out["sites"], "reads": sites_out["reads"],
                           "read_len": L, "graph_len": float(np.mean(sset.g_len)), "parallelism": "sites x%d" % world,
                           "sites_per_s": sites_out["sites_per_s"]},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                             "kernel": "pg_fill_kernel<%d, false>" % (2 * ((L + 31) // 32)),
                             "launches": int(tim["fill_launches"]),
                             "avg_launch_ms": tim["fill_ms"] / max(1, tim["fill_launches"]),
                             "alg_bytes_per_launch": b_alg_mine * args.steps / max(1, tim["fill_launches"]),
                             "note": "rank 0's shard (%d reads)" % reads_mine},
                "kernel_ms": {"fill": tim["fill_ms"], "trace": tim["trace_ms"]},
                "sites": sites_out, "dist": dist_info,
            }
    else:
        out = run_config2(args, env, ctx, capi, synth, site, arr, dist_info)
        if want_sites:
            log("config3 leg")
            sites_out, _, _, _ = run_sites_leg(args, env, ctx, capi, synth, sset, args.sites_steps, 1, False)
            if rank == 0:
                out["sites"] = sites_out
    rc = 0
    if rank == 0:
        print(json.dumps(out), flush=True)
        if out.get("verified") and out["verified"]["mismatches"]:
            rc = 3
        if out.get("sites") and out["sites"].get("reduce_equals_single") is False:
            rc = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def run_config2(args, env, ctx, capi, synth, site, arr, dist_info):
    torch, dist = env["torch"], env["dist"]
    rank, world, device = env["rank"], env["world"], env["device"]
    from paragraph_amd import dist as pgdist
    L = args.read_len
    G = site.total_len
    b_alg_total = args.reads * (6 * L * G + L + 64)
    graphs = ctx.upload_graphs([(site.seqs, site.edges)])
    graphs.set_labels([site.labels])
    # Two batch objects hold the same reads and take turns: consecutive steps then depend on nothing but the device, so the
    # traceback + count of step n (second stream) run under the fills of step n + 1 -- the steady state of a pipeline whose
    # batches are different reads.  Every step still does all of its work inside the timed region.
    batches = [ctx.new_batch(), ctx.new_batch()]
    batch = batches[0]
    packed = synth.packed_to_capi(arr)
    frag = np.arange(args.reads, dtype=np.uint32) // 2  # mates: reads 2k and 2k+1 form fragment k (ReadCounting.cpp:52-94)
    t0 = time.perf_counter()
    batch.upload(graphs, packed)
    batch.set_fragments(frag)
    ctx.sync()
    t_upload = time.perf_counter() - t0
    batches[1].upload(graphs, packed)
    batches[1].set_fragments(frag)
    ctx.sync()
    log("uploaded in %.2fs" % t_upload)

    # per-site counter table {count, READS, FWD, REV} x (nodes, edges, sequence sets) + filter tallies: a torch
    # tensor so that the reduce is ONE RCCL all-reduce over xGMI on device memory, no host hop
    n_counters = int(graphs.layout.n_counters)
    counts_t = torch.zeros(n_counters, dtype=torch.int32, device=device)

    step_no = [0]

    def step():
        b = batches[step_no[0] & 1]
        step_no[0] += 1
        ctx.counts_zero(counts_t.data_ptr(), n_counters)  # on the stream the count kernels run on: ordered with them
        b.align(capi.AF_ALL)
        b.count(remove_nonuniq=True, bad_align_frac=0.8, d_counts=counts_t.data_ptr())
        if world > 1:
            ctx.sync_compute()
            pgdist.allreduce_counts(counts_t)
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    env["barrier"]()
    log("warmup done")
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    env["barrier"]()
    elapsed = env["max_over_ranks"](time.perf_counter() - t0)
    log("timed region %.3fs" % elapsed)
    tim = ctx.timing()
    ctx.timing_enable(False)

    # results of the last timed step (PCIe-inclusive figure, verification)
    batch = batches[(step_no[0] - 1) & 1]
    t0 = time.perf_counter()
    res, ops = batch.download()
    t_download = time.perf_counter() - t0
    tab = counts_t.cpu().numpy().view(np.uint32)
    site_counts = capi.decode_counts(graphs, tab)[0]

    # PCIe-inclusive leg, streaming form: PINNED host arrays -> device -> results in pinned host arrays, two batch objects
    # and two sets of staging buffers; the upload of batch i+1 and the download of batch i-1 are DMAs on the copy stream
    # under the kernels of batch i
    t_stream = None
    if world == 1 and args.stream_batches > 0:
        n = args.reads
        n_ops = len(ops)
        bb, pins = [ctx.new_batch(), ctx.new_batch()], []
        for b in bb:
            pin = {"off": ctx.pinned_copy(packed[0]), "bases": ctx.pinned_copy(arr.reshape(-1)),
                   "gor": ctx.pinned_copy(np.zeros(n, np.uint32)), "frag": ctx.pinned_copy(frag)}
            b.upload(graphs, (pin["off"], pin["bases"]), pin["gor"])  # allocate once (steady state)
            b.set_fragments(pin["frag"])
            b.align(capi.AF_ALL)
            b.count(remove_nonuniq=True, bad_align_frac=0.8)
            _, _, p0 = b.download_counts()
            pin["res"] = ctx.pinned_empty(n, capi.RESULT_DTYPE)
            pin["ops"] = ctx.pinned_empty(n_ops + n_ops // 4 + 1024, np.uint32)
            pin["counts"] = ctx.pinned_empty(n_counters, np.uint32)
            pin["sup"] = ctx.pinned_empty(n, capi.SUPPORT_DTYPE)
            pin["path"] = ctx.pinned_empty(len(p0) + len(p0) // 4 + 1024, np.uint32)
            pins.append(pin)
        ctx.sync()

        def fetch(i):
            bb[i].download(into=(pins[i]["res"], pins[i]["ops"]))
            bb[i].download_counts(into=(pins[i]["counts"], pins[i]["sup"], pins[i]["path"]))

        t0 = time.perf_counter()
        pending = None
        for i in range(args.stream_batches):
            k = i & 1
            bb[k].upload(graphs, (pins[k]["off"], pins[k]["bases"]), pins[k]["gor"])
            bb[k].set_fragments(pins[k]["frag"])
            bb[k].align(capi.AF_ALL)
            bb[k].count(remove_nonuniq=True, bad_align_frac=0.8)
            if pending is not None:
                fetch(pending)
            pending = k
        fetch(pending)
        ctx.sync()
        t_stream = (time.perf_counter() - t0) / args.stream_batches
        log("streaming leg: %.3fs per batch" % t_stream)
        # same records as the resident pass (ops_off aside: CIGAR elements are bump-allocated in completion order)
        fields = [f for f in capi.RESULT_DTYPE.names if f != "ops_off"]
        stream_same = all(np.array_equal(pins[pending]["res"][:n][f], res[f]) for f in fields)
        for b in bb:
            b.close()
    if rank != 0:
        return None

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
            # PMC bytes (rocprofv3 FETCH_SIZE x measured 2.0 + WRITE_SIZE, separate passes) per read x the reads one
            # launch of THIS run processes
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
            "workload": "configs[1]: 1 DEL graph (200bp flanks, nodes 201/100/201), %d synthetic %dbp reads per GPU, "
                        "GraphAligner::alignRead(AF_ALL) = 4 fills + strand pick + traceback per read, then "
                        "filters + node/edge/sequence counts%s"
                        % (args.reads, L, " + 1 all-reduce of the counter table per step" if world > 1 else ""),
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
            "streaming_vs_value": (args.reads / t_stream / value) if t_stream else None,
            "streaming_results_equal_resident": stream_same if t_stream else None,
            "note": "upload_s / download_s: first upload (allocations included) and a download into fresh pageable arrays; "
                    "streaming = double-buffered steady state through pinned staging (pg_host_alloc): packed reads in, "
                    "results + ops + supports + paths + counts back on the host",
        },
        "counts": {"edges": {"%s_%s" % (site.names[a], site.names[b]): c[0] for (a, b), c in site_counts["edge_counts"].items()},
                   "sequences": {k: v[0] for k, v in site_counts["seq_counts"].items()},
                   "tallies": site_counts["tallies"],
                   "note": "fragment counts / read tallies of ONE step (the table is zeroed on the ctx stream per step), summed "
                           "over %d rank(s)" % world},
        "dist": dist_info,
    }
    if world == 1 and not args.no_cpu_baseline:
        log("cpu baseline leg")
        base, ref_res, ref_cig = run_cpu_leg(args, arr)
        out["cpu_baseline"] = base
        out["verified"] = verify_against_reference(capi, res, ops, ref_res, ref_cig)
    for b in batches:
        b.close()
    graphs.close()
    return out


def main():
    args = parse_args()
    if args.cpu_leg:
        cpu_leg_main(args)
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["PG_BENCH_LAUNCHER"] = "bench.py --gpus %d (self-spawned torch.distributed.run)" % args.gpus
        return spawn_ranks(args)
    return main_rank(args)


if __name__ == "__main__":
    sys.exit(main())
