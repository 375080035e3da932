#!/usr/bin/env python
"""bench.py -- headline benchmark of the read -> variant-graph realignment path on MI355X.

A "step" is one pass of the hot path (the default grmpy cascade GraphAligner::alignRead(AF_ALL) per read: the record the
reference reads off four graph fills, strand pick + traceback -- computed by the library's lean gssw stage from the three fills that
can change it, the fourth where it can; `plain_stage` runs the same steps with all four) over one batch of synthetic
reads that is already
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
  roofline     -- dominant kernel(s) (the lean stage's two fill launches per chunk taken together; pg_fill_kernel for the plain
                  stage): algorithmic bytes of SURVEY.md 8(d) (B_alg = 6*L*G + L + 64 per read) / HIP-event duration of the launches /
                  8 TB/s, and what measurably bounds them (VALU issue slots from SQ counters, HBM bytes from PMC counters)
  plain_stage  -- the same steps through the plain gssw stage (four fills per read), its rate and bounds, and the lean step's
                  records against its records (exit status 3 if any differ)
  cpu_baseline -- the reference's own gssw.c (oracle/_ref, kind "reference") or the plain-C port, timed on this
                  host's cores in a separate process (threads in one process AND one process per chunk; the better
                  one is `value`), N=1 only
  verified     -- the GPU results of the last timed step compared field by field and CIGAR by CIGAR with the
                  reference alignments the cpu_baseline leg computed for the same reads (exit status 3 on a mismatch)
  sites        -- the config3 leg (sites/s, strong scaling, reduce_equals_single)
  e2e          -- "sites genotyped/sec": BAM -> genotypes through the host workflow (pgw_genotype_graphs: BGZF / BAM decode, read
                  extraction, graph loading, device batches, count documents, genotyping, the JSON array written out), K timed
                  passes over a synthetic 10 000-site data set written before the timed region; CPU seconds per (site, sample),
                  genotypes against the simulated truth, a sample of sites' edge counts against the reference's own code on the
                  reads the reference's extraction policy keeps (exit status 3 on a mismatch)
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
# what the kernels compute in: two 16-bit lanes per VGPR (read / reverse complement), each an IEEE half holding the exact
# integer 1024 + score + frame (every integer below 2048 is exact in f16), i.e. integer DP carried by v_pk_*_f16 -- gssw's
# u8 (reads <= 250 bp) / i16 (251-512 bp) score semantics bit for bit, not a reduced-precision approximation
DTYPE = "f16x2 packed, exact integers < 2048 (gssw u8 / i16 score semantics)"
SIMDS = 256 * 4  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs (the clock and the cycles per instruction are measured: valu_bound)
CIGAR_STRIDE = 128
SITES_CIGAR_STRIDE = 256  # config-3 reads cross insertions of up to 1 000 bases


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
    ap.add_argument("--hot-site-depth", type=float, default=0.0,
                    help="config3: add one site sequenced this deep (e.g. 1500 = ~10 000 reads, grmpy's per-site cap)")
    ap.add_argument("--split-reads", type=int, default=5000,
                    help="config3 with N > 1: a site with this many reads or more is split over all ranks by fragment id")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--workspace-gib", type=float, default=165.0,
                    help="HBM budget for traceback state (two halves: trace of chunk i overlaps fill of chunk i+1); 165 GiB of the "
                         "288 GB = 2 fill launches of 500 k reads per 1 M-read step (128: 3 of 333 k, 1.2 %% slower; 64: 5 of 200 k, "
                         "2.6 %% slower: every launch ends with a tail in which the chip drains, profiles/r05_headline_workspace_ab.jsonl)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-batches", type=int, default=16,
                    help="batches of the PCIe-inclusive streaming leg (0 = skip; reported beside the headline value)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_r06.json"),
                    help="PMC-derived HBM bytes per fill launch (written by tools/pmc_traffic.py); used only when the kernel "
                         "sources it was collected on are the ones of this build (kernel_source_sha)")
    ap.add_argument("--sq-json", default=os.path.join(ROOT, "profiles", "r06_sq_counters.json"),
                    help="SQ counters of one fill launch (tools/sq_collect.sh + tools/sq_summary.py), same rule")
    ap.add_argument("--traffic-json-plain", default=os.path.join(ROOT, "profiles", "traffic_r06_plain_stage.json"),
                    help="the same for the PLAIN gssw stage's fill kernel (the `plain_stage` leg; collected with PG_LEAN=0)")
    ap.add_argument("--sq-json-plain", default=os.path.join(ROOT, "profiles", "r06_sq_counters_plain_stage.json"),
                    help="SQ counters of the plain stage's fill launch, same rule")
    ap.add_argument("--isa-mix-json", default=os.path.join(ROOT, "profiles", "r06_fill_isa_mix.json"),
                    help="static VALU mix of a step by issue class (tools/isa_mix.py), same rule")
    ap.add_argument("--valu-rate-json", default=os.path.join(ROOT, "profiles", "r04_valu_rate.json"),
                    help="measured cycles per wave64 instruction (tools/ubench/valu_rate)")
    ap.add_argument("--clock-json", default=os.path.join(ROOT, "profiles", "r04_clock_probe.json"),
                    help="engine clock sampled under the fill's load (tools/clock_probe.py)")
    ap.add_argument("--collective", default="auto", choices=["auto", "on", "off"],
                    help="the all-reduce of the counter table inside every step.  auto/on: always -- with ONE rank a world-size-1 "
                         "RCCL communicator is created, so N = 1 runs the code path of N = 8 (auto falls back to off, and says "
                         "so in `dist`, if the communicator cannot be created); off: only with more than one rank")
    # internal: the CPU baseline runs in its own process (it forks workers; the GPU process must not)
    ap.add_argument("--sites-verify", type=int, default=500,
                    help="config3 leg: sites of rank 0's shard whose alignments, per-read outcome and count tables are compared "
                         "with the reference's code in a CPU-leg process (0 = skip); exit status 3 on a mismatch")
    ap.add_argument("--exact-shortcut-steps", type=int, default=3,
                    help="timed steps of the reported-only leg `exact_shortcut` (the headline workload with pg_batch_retire_exact_matches "
                         "in front of the gssw stage: reads whose alignRead record one exact full-length match forces skip their "
                         "fills; records and count table compared with the plain step's); 0 = leave it out")
    ap.add_argument("--plain-steps", type=int, default=2,
                    help="timed steps of the reported-only leg `plain_stage` (the headline workload through the plain gssw stage, four fills "
                         "per read, and the lean step's records against its records); 0 = leave it out")
    ap.add_argument("--shortcut-in-process", action="store_true", help=argparse.SUPPRESS)  # (the child of the exact_shortcut leg)
    ap.add_argument("--no-e2e-shortcut", action="store_true", help="leave the e2e leg's `with_exact_shortcut` passes out (A/B of the legs behind them)")
    ap.add_argument("--e2e-steps", type=int, default=3,
                    help="timed passes of the BAM -> genotypes leg (0 = skip): every pass takes ALL sites of the e2e data set from the "
                         "BAM file to genotype documents written as JSON (with N ranks: rank r takes sites r, r + N, ...)")
    ap.add_argument("--e2e-sites", type=int, default=10000, help="sites of the e2e data set (one 30x sample; the whole job)")
    ap.add_argument("--e2e-verify", type=int, default=500,
                    help="e2e leg: sites of rank 0's shard whose edge counts are compared with the reference's code (0 = skip)")
    ap.add_argument("--e2e-threads", type=int, default=0, help="host threads of the workflow per rank (0 = usable CPUs / ranks)")
    ap.add_argument("--e2e-options", default="", help="JSON object of further pgw_genotype_graphs options for the e2e leg (A/B runs: "
                                                       "sites_per_batch, lanes, ...)")
    ap.add_argument("--config5-graphs", type=int, default=100,
                    help="configs[4] leg of the default run: graphs with one 2-8 kb inline ALT node each (0 = skip)")
    ap.add_argument("--config5-reads-per-graph", type=int, default=240, help="250 bp reads per graph of that leg")
    ap.add_argument("--config5-verify-per-graph", type=int, default=8,
                    help="reads of every graph of that leg compared with the reference's gssw.c in a CPU-leg process (0 = skip)")
    ap.add_argument("--config5-steps", type=int, default=3, help="timed passes of that leg")
    ap.add_argument("--cpu-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--config5-cpu-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--sites-cpu-leg", action="store_true", help=argparse.SUPPRESS)
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


def _cpu_throttle():
    """(periods, throttled periods, throttled microseconds) of this process's cgroup so far (cpu.stat of cgroup v2), or None: a
    leg that wants more CPU than the quota in some 100 ms period is stopped for the rest of it -- the e2e leg reports how often."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            kv = dict(line.split() for line in f if len(line.split()) == 2)
        return int(kv["nr_periods"]), int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except (OSError, ValueError, KeyError):
        return None


def _throttle_delta(a, b):
    if a is None or b is None:
        return None
    return {"cgroup_periods": b[0] - a[0], "throttled_periods": b[1] - a[1], "throttled_ms": (b[2] - a[2]) / 1e3}


def _effective_cpus():
    """CPUs this process can really use at once: its affinity mask, cut to the cgroup's quota (a container on a 256-CPU host
    may be held to 16: forking one generator per visible CPU in each of 8 ranks would start 2 048 processes on 16 cores)."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cpu_quota()
    if quota:
        ncpu = min(ncpu, max(1, int(quota + 0.5)))
    return max(1, ncpu)


def reference_site_outcome(chk, s, stride=256):
    """The reference's outcome for ONE config-3 site (TEST INFRASTRUCTURE: runs in the CPU-leg process / tests only): every
    read aligned by the checker (the reference's gssw.c where oracle/_ref is built), then the reference's filters,
    disambiguation and counting (graph-tools compiled as it lies + the Disambiguation.cpp / ReadCounting.cpp glue RESTATED in
    oracle/ref_counts.cpp -> oracle/_ref/libpg_refcounts.so, or the Python restatement): alignments, CIGAR slots, per-read status and label sets, node / edge / sequence tables."""
    from oracle import counts as oc
    from oracle import oracle as orc
    arr = s.reads
    n, L = arr.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
    res = np.zeros(n, dtype=orc.RESULT_NP)
    cig = np.zeros((n, stride), dtype=np.uint8)
    if n:
        chk.align_into(s.site.seqs, s.site.edges, off, np.ascontiguousarray(arr).reshape(-1), res, cig, threads=1)
    recs = [{"pos": int(r["graph_pos"]), "cigar": bytes(c).split(b"\0", 1)[0].decode(), "aligned": int(r["score"]) > 0,
             "unique": bool(r["unique"]), "graph_reverse": bool(s.is_reverse[k]) != bool(r["returned_reverse"]), "read_len": L,
             "fragment": int(s.fragment[k])} for k, (r, c) in enumerate(zip(res, cig))]
    labels = sorted({l for v in s.site.labels.values() for l in v})
    from oracle import select
    count = select.count_site()
    wc = count(oc.CountGraph(s.site.seqs, s.site.edges, s.site.labels, labels), recs, remove_nonuniq=True)
    lab_idx = {l: k for k, l in enumerate(labels)}
    return {"res": res, "cig": cig, "status": np.array(wc["status"], dtype=np.uint8),
            "label_mask": np.array([sum(1 << lab_idx[l] for l in ls) for ls in wc["labels"]], dtype=np.uint64),
            "nodes": wc["nodes"], "edges": wc["edges"], "node_counts": np.asarray(wc["node_counts"], dtype=np.uint64),
            "edge_counts": np.asarray(wc["edge_counts"], dtype=np.uint64), "seq_counts": wc["seq_counts"]}


_SITES_JOB = None


def _sites_job(i):
    from oracle import select
    return reference_site_outcome(select.gssw(), _SITES_JOB[i], SITES_CIGAR_STRIDE)


def sites_cpu_leg_main(args):
    """CPU-leg process of the config-3 leg: the reference's outcome for a sample of sites (pickled SiteReads in, per-site
    dicts out), fanned out over the CPUs the cgroup allows."""
    global _SITES_JOB
    import multiprocessing as mp
    import pickle
    from oracle import counts as oc
    from oracle import oracle as orc
    with open(args.cpu_reads_file, "rb") as f:
        _SITES_JOB = pickle.load(f)
    procs = max(1, min(_effective_cpus(), len(_SITES_JOB)))
    t0 = time.perf_counter()
    if procs == 1:
        out = [_sites_job(i) for i in range(len(_SITES_JOB))]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            out = pool.map(_sites_job, range(len(_SITES_JOB)), chunksize=max(1, len(_SITES_JOB) // (4 * procs)))
    with open(args.cpu_out, "wb") as f:
        pickle.dump(out, f, protocol=4)
    print(json.dumps({"sites": len(out), "reads": int(sum(len(o["res"]) for o in out)), "seconds": time.perf_counter() - t0, "workers": procs,
                      "aligner": "reference gssw.c (oracle/_ref)" if orc.have_ref() else "plain-C restatement (oracle/pg_oracle.c)",
                      "counting": "graph-tools compiled + Disambiguation glue restated (oracle/_ref/libpg_refcounts.so)" if oc.have_ref()
                      else "restatement (oracle/counts.py)"}))


def run_sites_cpu_leg(sites):
    """Runs sites_cpu_leg_main in a fresh interpreter (it forks; this process holds a HIP context): (info, per-site dicts)."""
    import pickle
    tmp = tempfile.mkdtemp(prefix="pgbench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    in_file, out_file = os.path.join(tmp, "sites.pkl"), os.path.join(tmp, "ref.pkl")
    try:
        with open(in_file, "wb") as f:
            pickle.dump(sites, f, protocol=4)
        cmd = [sys.executable, os.path.abspath(__file__), "--sites-cpu-leg", "--cpu-reads-file", in_file, "--cpu-out", out_file]
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, check=True)
        info = json.loads(p.stdout.decode().strip().splitlines()[-1])
        with open(out_file, "rb") as f:
            return info, pickle.load(f)
    finally:
        for f in (in_file, out_file):
            if os.path.exists(f):
                os.unlink(f)
        os.rmdir(tmp)


def verify_sites(capi, graphs, sample_sites, sample_idx, res, ops, sup, table, want, info):
    """The sample sites' GPU outcome (first reads of the rank's batch, in sample order) against the reference's: alignments +
    CIGARs, per-read status of the count path and label sets of the mapped reads, node / edge / sequence tables per site."""
    n = int(sum(len(s.reads) for s in sample_sites))
    ref_res = np.concatenate([w["res"] for w in want]) if want else np.zeros(0)
    ref_cig = np.concatenate([w["cig"] for w in want]) if want else np.zeros((0, SITES_CIGAR_STRIDE), np.uint8)
    out = verify_against_reference(capi, res[:n], ops, ref_res, ref_cig)
    ref_status = np.concatenate([w["status"] for w in want])
    ref_labels = np.concatenate([w["label_mask"] for w in want])
    bad_status = sup["status"][:n] != ref_status
    mapped = ref_status == 1
    bad_labels = (sup["label_mask"][:n] != ref_labels) & mapped
    cnt = capi.decode_counts(graphs, table)
    bad_tables = 0
    first_bad_site = None
    for s, si, w in zip(sample_sites, sample_idx, want):
        c = cnt[int(si)]
        ok = np.array_equal(c["node_counts"], w["node_counts"]) and c["seq_counts"] == w["seq_counts"] and \
            all(c["edge_counts"][tuple(e)] == [int(x) for x in w["edge_counts"][ei]] for ei, e in enumerate(s.site.edges))
        if not ok:
            bad_tables += 1
            if first_bad_site is None:
                first_bad_site = int(si)
    out.update({"sites": len(sample_sites), "status_mismatches": int(bad_status.sum()), "label_set_mismatches": int(bad_labels.sum()),
                "site_table_mismatches": int(bad_tables), "first_bad_site": first_bad_site,
                "mismatches": int(out["mismatches"]) + int(bad_status.sum()) + int(bad_labels.sum()) + int(bad_tables),
                "fields": out["fields"] + "; count-path status of every read, label sets of the mapped reads; node / edge / sequence "
                                          "tables of every sampled site (after the all-reduce)",
                "reference": info})
    return out


# ---------------------------------------------------------------------------------------------------
# configs[4]: long inline ALT nodes (2-8 kb) + 250 bp reads
# ---------------------------------------------------------------------------------------------------
CONFIG5_CIGAR_STRIDE = 256
CONFIG5_READ_LEN = 250


def config5_cases(n_graphs, reads_per_graph, seed=5):
    """INV / DUP-style graphs LF -> {REF (60 bp), ALT (inline sequence, 2 000 .. 8 000 bp)} -> RF with 300 bp flanks, 250 bp
    reads from both haplotypes (1 % substitutions, 1 % of the reads with an indel, 0.5 % unrelated sequence)."""
    from paragraph_amd import synth
    out = []
    for gi in range(n_graphs):
        alt = 2000 + (gi * 6007) % 6001
        site = synth.long_node_site(seed * 1000 + gi, alt)
        arr = synth.simulate_reads_packed(site, reads_per_graph, CONFIG5_READ_LEN, seed * 2000 + gi, indel_frac=0.01, random_frac=0.005)
        out.append((site, arr))
    return out


_C5_JOB = None


def _c5_job(i):
    from oracle import oracle as orc
    from oracle import select
    site, arr = _C5_JOB[i]
    n, L = arr.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
    res = np.zeros(n, dtype=orc.RESULT_NP)
    cig = np.zeros((n, CONFIG5_CIGAR_STRIDE), dtype=np.uint8)
    select.gssw().align_into(site.seqs, site.edges, off, np.ascontiguousarray(arr).reshape(-1), res, cig, threads=1)
    return res, cig


def config5_cpu_leg_main(args):
    """CPU-leg process of the configs[4] leg (TEST INFRASTRUCTURE): the reference's gssw.c on the sampled reads of every graph."""
    global _C5_JOB
    import multiprocessing as mp
    import pickle
    from oracle import oracle as orc
    with open(args.cpu_reads_file, "rb") as f:
        _C5_JOB = pickle.load(f)
    procs = max(1, min(_effective_cpus(), len(_C5_JOB)))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        out = pool.map(_c5_job, range(len(_C5_JOB)), chunksize=max(1, len(_C5_JOB) // (4 * procs)))
    spent = time.perf_counter() - t0
    with open(args.cpu_out, "wb") as f:
        pickle.dump(out, f, protocol=4)
    n = int(sum(len(r) for r, _ in out))
    print(json.dumps({"graphs": len(out), "reads": n, "seconds": spent, "workers": procs, "reads_per_s": n / max(spent, 1e-9),
                      "aligner": "reference gssw.c (oracle/_ref)" if orc.have_ref() else "plain-C restatement (oracle/pg_oracle.c)"}))


def run_config5_leg(args, env, ctx, capi):
    """configs[4] on the device: every graph's reads aligned (4 fills each over 2.7 - 8.7 k columns, the long node swept in one go:
    a lane's state crosses a column block boundary in registers) + counted, K timed passes on resident data; a sample of every
    graph's reads against the reference's gssw.c."""
    import pickle
    rank = env["rank"]
    cases = config5_cases(args.config5_graphs, args.config5_reads_per_graph)
    L = CONFIG5_READ_LEN
    graphs = ctx.upload_graphs([(s.seqs, s.edges) for s, _ in cases])
    graphs.set_labels([s.labels for s, _ in cases])
    arr = np.concatenate([a for _, a in cases])
    n = len(arr)
    gor = np.repeat(np.arange(len(cases), dtype=np.uint32), args.config5_reads_per_graph)
    g_len = np.array([s.total_len for s, _ in cases], dtype=np.float64)
    cells = float((4.0 * L * g_len * args.config5_reads_per_graph).sum())
    b_alg = float(((6.0 * L * g_len + L + 64) * args.config5_reads_per_graph).sum())
    b_alg_h = float(((2.0 * L * g_len + L + 64) * args.config5_reads_per_graph).sum())
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
    batches = [ctx.new_batch(), ctx.new_batch()]
    for b in batches:
        b.upload(graphs, (off, np.ascontiguousarray(arr).reshape(-1)), gor)
        b.set_fragments(np.arange(n, dtype=np.uint32) // 2)
    ctx.sync()

    def one(k):
        batches[k & 1].align(capi.AF_ALL)
        batches[k & 1].count(remove_nonuniq=True, bad_align_frac=0.8)

    one(0)
    one(1)
    env["barrier"]()
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for k in range(args.config5_steps):
        one(k)
    env["barrier"]()
    elapsed = env["max_over_ranks"](time.perf_counter() - t0)
    tim = ctx.timing()
    ctx.timing_enable(False)
    last = batches[(args.config5_steps - 1) & 1]
    res, ops = last.download()
    fill_s = tim["fill_ms"] / 1e3
    out = {"workload": "configs[4]: %d graphs LF(300) -> {REF(60), ALT(2 000 - 8 000, inline)} -> RF(300), %d synthetic 250 bp reads each, "
                       "alignRead(AF_ALL) + filters + counts, every rank its own copy" % (len(cases), args.config5_reads_per_graph),
           "graphs": len(cases), "reads": n, "read_len": L, "mean_graph_len": float(g_len.mean()), "steps": args.config5_steps,
           "ms_per_step": elapsed / args.config5_steps * 1e3, "reads_per_s": n * env["world"] * args.config5_steps / elapsed,
           "cell_updates_per_s": cells * env["world"] * args.config5_steps / elapsed,
           "fill_launches": int(tim["fill_launches"]), "fill_ms": tim["fill_ms"], "trace_ms": tim["trace_ms"],
           "hbm_formula_frac": (b_alg * args.config5_steps / fill_s / 1e9 / HBM_PEAK_GBS) if fill_s > 0 else None,
           "hbm_alg_h_only_frac": (b_alg_h * args.config5_steps / fill_s / 1e9 / HBM_PEAK_GBS) if fill_s > 0 else None}
    v = args.config5_verify_per_graph
    if rank == 0 and v > 0:
        sample = [(s, a[:v]) for s, a in cases]
        tmp = tempfile.mkdtemp(prefix="pgbench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        in_file, out_file = os.path.join(tmp, "c5.pkl"), os.path.join(tmp, "ref.pkl")
        try:
            with open(in_file, "wb") as f:
                pickle.dump(sample, f, protocol=4)
            cmd = [sys.executable, os.path.abspath(__file__), "--config5-cpu-leg", "--cpu-reads-file", in_file, "--cpu-out", out_file]
            cenv = dict(os.environ)
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                cenv.pop(k, None)
            p = subprocess.run(cmd, stdout=subprocess.PIPE, env=cenv, check=True)
            info = json.loads(p.stdout.decode().strip().splitlines()[-1])
            with open(out_file, "rb") as f:
                want = pickle.load(f)
        finally:
            for f in (in_file, out_file):
                if os.path.exists(f):
                    os.unlink(f)
            os.rmdir(tmp)
        idx = (np.arange(len(cases))[:, None] * args.config5_reads_per_graph + np.arange(v)[None, :]).reshape(-1)
        ref_res = np.concatenate([w[0] for w in want])
        ref_cig = np.concatenate([w[1] for w in want])
        ver = verify_against_reference(capi, res[idx], ops, ref_res, ref_cig)
        ver["reference"] = info
        ver["sample"] = "the first %d reads of every graph" % v
        out["verified"] = ver
    for b in batches:
        b.close()
    graphs.close()
    return out


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
    from oracle import select
    chk = select.gssw()  # PG_REQUIRE_REF=1: no silent change of checker
    kind, label = ("reference", "reference gssw.c (oracle/_ref)") if orc.have_ref() else ("port", "plain-C restatement (oracle/pg_oracle.c)")
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
    # (the lean stage: the forward-graph fill of the strand that was not returned may not have run -- multi_mask bit 4 says so and that
    #  fill's bit reads 0; nothing of the reference's Read depends on it)
    skipped_other = (g["multi_mask"] & 0x10) != 0
    for k in range(4):
        not_run = skipped_other & ((g["returned_reverse"] != 0) != (k == 1)) if k < 2 else np.zeros(n, dtype=bool)
        bad |= (((g["multi_mask"] >> k) & 1).astype(np.int32) != ref_res["multi"][:, k]) & ~not_run
    gc = capi.render_cigars(g, ops, ref_cig.shape[1])
    bad |= (gc != ref_cig).any(axis=1)
    first = int(np.nonzero(bad)[0][0]) if bad.any() else None
    out = {"reads": int(n), "mismatches": int(bad.sum()),
           "fields": "graph_pos, score, mapq, unique, returned_reverse, multi[4] (of the fills that ran), CIGAR string",
           "forward_fills_of_the_other_strand_not_run": int(skipped_other.sum())}
    if first is not None:
        out["first_mismatch"] = {"read": first, "gpu_cigar": bytes(gc[first]).split(b"\0")[0].decode(),
                                 "ref_cigar": bytes(ref_cig[first]).split(b"\0")[0].decode(),
                                 "gpu": [int(g[first][f]) for f in ("graph_pos", "score", "mapq", "is_unique", "returned_reverse", "multi_mask", "status")],
                                 "ref": [int(ref_res[first][f]) for f in ("graph_pos", "score", "mapq", "unique", "returned_reverse")]}
    return out


# ---------------------------------------------------------------------------------------------------
# bounds that bind: measured HBM traffic and VALU issue of the fill kernel (rocprofv3 --pmc, collected separately)
# ---------------------------------------------------------------------------------------------------
def _counter_file(path, sha):
    """A counter file is evidence for THIS run only if it was collected on the kernel sources of this build."""
    if not path or not os.path.exists(path):
        return None, {"file": None, "usable": False, "why": "no counter file"}
    try:
        with open(path) as f:
            doc = json.load(f)
    except Exception as e:  # noqa: BLE001
        return None, {"file": os.path.relpath(path, ROOT), "usable": False, "why": "unreadable: %s" % e}
    src = {"file": os.path.relpath(path, ROOT), "kernel_source_sha": doc.get("kernel_source_sha"),
           "collected_at_head": doc.get("collected_at_head"), "reads_in_pmc_run": doc.get("reads_in_pmc_run"),
           "this_build_kernel_source_sha": sha, "usable": doc.get("kernel_source_sha") == sha}
    if not src["usable"]:
        src["why"] = "collected on other kernel sources: not reported as this run's"
        return None, src
    return doc, src


def _class_cycles(rate_doc):
    """cycles per wave64 instruction per SIMD of the two issue classes, from the microbenchmark's rows at 8 wavefronts per SIMD
    and 8 independent chains (the issue rate) x the clock s_memtime ran at in the same rows"""
    rows = [r for r in rate_doc["rows"] if r["waves_per_simd"] == 8 and r["chains"] == 8]
    ghz = {r["op"]: r["memtime_per_memrealtime_median"] * 0.1 for r in rows}  # s_memrealtime counts at 100 MHz
    cyc = {r["op"]: r["ns_per_wave_inst_per_simd"] * ghz[r["op"]] for r in rows}
    packed = [cyc[k] for k in ("pk_maximum3_f16", "pk_max_u16", "pk_add_f16", "perm_b32", "bfi_b32", "mov_dpp_row_shr1") if k in cyc]
    plain = [cyc[k] for k in ("add_u32", "add_u32_literal", "and_b32", "mov_b32") if k in cyc]
    return sum(packed) / len(packed), sum(plain) / len(plain)


def valu_bound(args, sha, reads_per_launch, avg_launch_s):
    """The share of the SIMDs' VALU issue cycles the fill launches of this run used, from MEASURED constants:
      cycles per wave64 instruction of the two issue classes   tools/ubench/valu_rate (profiles/r04_valu_rate.json): packed /
                                                               VOP3 / DPP / SGPR-operand instructions 4.2, plain 32-bit add /
                                                               logic / mov 2.3
      sustained engine clock under the fill                    amd-smi samples during a 60-step bench (profiles/r04_clock_probe.json)
      VALU instructions per wave-step                          SQ_INSTS_VALU of one launch (dynamic, rare paths included)
      their split by issue class                               static, common path of a step of each graph direction
                                                               (tools/isa_mix.py); what the dynamic count holds beyond the
                                                               common path (node boundaries) is priced as four-cycle
    The counter file and the static mix are used only for the kernel sources they were made on (kernel_source_sha)."""
    sq, src = _counter_file(args.sq_json, sha)
    mix, mix_src = _counter_file(args.isa_mix_json, sha)
    out = {"source": src, "isa_mix_source": mix_src}
    try:
        with open(args.valu_rate_json) as f:
            c4, c2 = _class_cycles(json.load(f))
        with open(args.clock_json) as f:
            clock_ghz = json.load(f)["under_fill_load"]["xcd_clock_mhz_median"] / 1e3
    except Exception as e:  # noqa: BLE001
        out["why"] = "no measured issue rates / clock: %s" % e
        return out
    out.update({"cycles_per_packed_inst": c4, "cycles_per_plain_inst": c2, "clock_ghz": clock_ghz, "simds": SIMDS,
                "rates_from": os.path.relpath(args.valu_rate_json, ROOT), "clock_from": os.path.relpath(args.clock_json, ROOT)})
    if not (sq and mix and avg_launch_s > 0):
        return out
    n = sq["reads_in_pmc_run"]
    dyn = sq["per_wave_step"]["SQ_INSTS_VALU"]
    dirs = mix["directions"]
    n4 = sum(d["per_step"].get("valu4", 0.0) for d in dirs.values()) / len(dirs)
    n2 = sum(d["per_step"].get("valu2", 0.0) for d in dirs.values()) / len(dirs)
    cyc_step = n4 * c4 + n2 * c2 + max(0.0, dyn - n4 - n2) * c4
    wave_steps = sq["wave_steps"] / n * reads_per_launch
    floor_s = cyc_step * wave_steps / (SIMDS * clock_ghz * 1e9)
    out.update({"insts_per_wave_step": dyn, "common_path_packed_per_step": n4, "common_path_plain_per_step": n2,
                "issue_cycles_per_wave_step": cyc_step, "issue_floor_ms": floor_s * 1e3, "issue_frac": floor_s / avg_launch_s,
                "note": "issue_frac = VALU issue cycles of the launch / (1 024 SIMDs x the measured clock x the launch's duration).  A "
                        "two-cycle instruction only costs two when another wavefront's two-cycle instruction pairs with it "
                        "(profiles/r04_iadd_ab.jsonl: turning 30 of 92 instructions per step from four-cycle into two-cycle ones "
                        "bought 3.5 %, not 18 %), so the true occupancy of the issue port lies between this figure and the one "
                        "with every instruction priced at the packed rate: issue_frac_all_packed",
                "issue_frac_all_packed": dyn * c4 * wave_steps / (SIMDS * clock_ghz * 1e9) / avg_launch_s})
    pairs = sq["per_wave_step"].get("SQ_ACTIVE_INST_VALU2")
    if pairs is not None:
        # SQ_ACTIVE_INST_VALU2 = quad-cycles in which the SIMD issued TWO VALU instructions: every instruction takes one quad-cycle
        # slot of its SIMD except those pairs, which share one
        out["dual_issue_quads_per_wave_step"] = pairs
        out["issue_frac_with_measured_pairing"] = (dyn - pairs) * 4.0 * wave_steps / (SIMDS * clock_ghz * 1e9) / avg_launch_s
    return out


def measured_bounds(args, reads_per_launch, avg_launch_s):
    """roofline fields beyond the contract's formula: what physically bounds pg_fill_kernel.
      traffic            HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated)
      hbm_measured_*     traffic / this run's average launch duration, against the 8 TB/s peak
      valu               SQ counters of one launch: VALU instructions per wave-step, cycles per instruction, and the share of
                         the SIMDs' issue cycles this run's launch duration leaves used (insts x cycles / (SIMDs x clock x t))"""
    from paragraph_amd import build as pgbuild
    sha = pgbuild.kernel_source_sha()
    out = {"traffic": None, "hbm_measured_gbs": None, "hbm_measured_frac": None, "valu": None}
    doc, src = _counter_file(args.traffic_json, sha)
    out["traffic_source"] = src
    if doc and avg_launch_s > 0:
        out["traffic"] = doc["hbm_bytes_per_read"] * reads_per_launch
        out["hbm_measured_gbs"] = out["traffic"] / avg_launch_s / 1e9
        out["hbm_measured_frac"] = out["hbm_measured_gbs"] / HBM_PEAK_GBS
    out["valu"] = valu_bound(args, sha, reads_per_launch, avg_launch_s)
    return out


def roofline_head(extra, formula_gbs):
    """The contract's six roofline fields for pg_fill_kernel, named after the bound that binds.

    The kernel is VALU-issue-bound: with SQ counters collected on THIS build's kernel sources `bound` is "valu",
    achieved / peak are wave64 VALU issue slots per second (one slot per SIMD per four cycles; a dual-issued pair shares one) and
    frac is the share of the issue port the launches held.  The two HBM figures stay beside it under their own names:
      hbm_formula_frac   SURVEY 8(d)'s B_alg (H, E and F of the two traced fills) / launch time / 8 TB/s.  The kernel keeps H only
                         and re-derives E / F in the traceback, so it does NOT move these bytes: the figure can exceed 1 and is
                         void as a fraction of peak -- kept because the contract defines it
      hbm_measured_frac  PMC bytes per launch / launch time / 8 TB/s: what the kernel really moves
    Without usable counters (kernel sources changed since they were collected) the line falls back to the HBM formula figure and
    says so."""
    valu = extra.get("valu") or {}
    head = {"traffic": extra.pop("traffic"), "hbm_formula_gbs": formula_gbs, "hbm_formula_frac": formula_gbs / HBM_PEAK_GBS,
            "hbm_formula_note": "SURVEY 8(d) formula bytes (H + E + F of two fills); void as a fraction for an H-only trace"}
    frac = valu.get("issue_frac_with_measured_pairing", valu.get("issue_frac"))
    if frac is not None:
        peak = SIMDS * valu["clock_ghz"] * 1e9 / 4.0 / 1e9
        head.update({"bound": "valu", "achieved": frac * peak, "peak": peak, "unit": "G wave64 VALU issue slots/s", "frac": frac,
                     "bound_note": "share of the 1 024 SIMDs' VALU issue slots (one per four cycles at the measured clock) the fill "
                                   "launches held; instructions per wave-step and dual-issue pairs from SQ counters of this build"})
    else:
        head.update({"bound": "hbm", "achieved": formula_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": formula_gbs / HBM_PEAK_GBS,
                     "bound_note": "no SQ counters for this build's kernel sources: the contract's HBM formula figure stands in; the "
                                   "kernel is VALU-issue-bound (see valu / hbm_measured_frac when counters are present)"})
    return head


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
    global layout on every rank and the all-reduce needs no index translation) and the reads of its own shard.  Whole sites
    go to one rank (dist.partition_sites); a HOT site -- `split_reads` reads or more; grmpy caps a site at 10 000,
    src/c++/main/grmpy.cpp:67 -- is split over all ranks by fragment id (dist.partition_fragments), never by read: counts are
    per fragment with mate-union semantics (ReadCounting.cpp:52-94, Fragment.cpp:141-181), so mates stay on one rank and the
    all-reduce then really sums that site's counters."""

    def __init__(self, sites, read_len, world, split_reads=5000):
        from paragraph_amd import dist as pgdist
        self.sites = sites
        self.L = read_len
        self.world = world
        self.n_reads_site = np.array([len(s.reads) for s in sites], dtype=np.int64)
        self.g_len = np.array([s.site.total_len for s in sites], dtype=np.int64)
        self.weights = self.n_reads_site * read_len * self.g_len  # DP cells per site
        self.hot = [int(i) for i in np.nonzero(self.n_reads_site >= split_reads)[0]] if world > 1 else []
        w = self.weights.copy()
        w[self.hot] = 0
        whole = np.array([i for i in range(len(sites)) if i not in set(self.hot)], dtype=np.int64)
        self.parts = [whole[p] for p in pgdist.partition_sites(w[whole], world)] if len(whole) else [whole] * world
        self.hot_reads = {i: pgdist.partition_fragments(sites[i].fragment, world) for i in self.hot}

    def rank_arrays(self, rank):
        """reads, graph of read, fragment id, strand flag of one rank: its whole sites + its fragments of the hot sites"""
        pieces = [(i, None) for i in self.parts[rank]] + [(i, self.hot_reads[i][rank]) for i in self.hot]
        return self._gather(pieces)

    def all_arrays(self):
        return self._gather([(i, None) for i in range(len(self.sites))])

    def _gather(self, pieces):
        arr, gor, frag, rev = [], [], [], []
        for i, sel in pieces:
            s = self.sites[i]
            sel = slice(None) if sel is None else sel
            r = s.reads[sel]
            arr.append(r)
            gor.append(np.full(len(r), i, dtype=np.uint32))
            frag.append(np.asarray(s.fragment)[sel])
            rev.append(np.asarray(s.is_reverse)[sel])
        if not arr:
            return np.zeros((0, self.L), np.uint8), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8)
        return (np.concatenate(arr), np.concatenate(gor), np.concatenate(frag).astype(np.uint32),
                np.concatenate(rev).astype(np.uint8))

    def rank_weight(self, rank):
        return float(self.weights[self.parts[rank]].sum()
                     + sum(len(self.hot_reads[i][rank]) * self.L * int(self.g_len[i]) for i in self.hot))

    def rank_reads(self, rank):
        return int(self.n_reads_site[self.parts[rank]].sum() + sum(len(self.hot_reads[i][rank]) for i in self.hot))

    def b_alg(self, rank):
        b = sum(int(self.n_reads_site[i]) * (6 * self.L * int(self.g_len[i]) + self.L + 64) for i in self.parts[rank])
        b += sum(len(self.hot_reads[i][rank]) * (6 * self.L * int(self.g_len[i]) + self.L + 64) for i in self.hot)
        return int(b)


def run_sites_leg(args, env, ctx, capi, synth, sset, steps, warmup, timed_events):
    """K timed passes over the rank's shard of the site set + ONE all-reduce of the table per pass.  Returns a dict
    (rank 0 adds the check against the 1-rank table)."""
    torch, dist = env["torch"], env["dist"]
    rank, world = env["rank"], env["world"]
    from paragraph_amd import dist as pgdist
    graphs = ctx.upload_graphs([(s.site.seqs, s.site.edges) for s in sset.sites])
    graphs.set_labels([s.site.labels for s in sset.sites])
    n_counters = int(graphs.layout.n_counters)
    arr, gor, frag, rev = sset.rank_arrays(rank)
    batch = ctx.new_batch()
    batch.upload(graphs, synth.packed_to_capi(arr), gor)
    batch.set_fragments(frag, rev)
    tables = [torch.zeros(n_counters, dtype=torch.int32, device=env["device"]) for _ in range(2)]
    red = env["reducer"]
    ctx.sync()
    torch.cuda.synchronize()  # torch.zeros ran on torch's stream, the library zeroes and counts on its own
    pass_no = [0]

    def one_pass(b, red=red):
        # stream-ordered: zero + fills + traceback + count are queued on the library's streams, the reduce behind an event on
        # a stream of torch's, the next pass (other table) right away -- the host waits for nothing until the barrier
        t = tables[pass_no[0] & 1]
        pass_no[0] += 1
        if red:
            red.acquire(t)
        ctx.counts_zero(t.data_ptr(), n_counters)
        b.align(capi.AF_ALL)
        b.count(remove_nonuniq=True, bad_align_frac=0.8, d_counts=t.data_ptr())
        if red:
            red.reduce(t)  # the only collective of the path
        return t

    for _ in range(warmup):
        one_pass(batch)
    env["barrier"]()
    if timed_events:
        ctx.timing_enable(True)
        ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        table = one_pass(batch)
    env["barrier"]()
    elapsed_mine = time.perf_counter() - t0
    elapsed = env["max_over_ranks"](elapsed_mine)
    tim = None
    if timed_events:
        tim = ctx.timing()
        ctx.timing_enable(False)
    # every rank's own view (a straggler GPU or an unbalanced shard shows here, not in the max)
    mine = {"rank": rank, "device": env["device"].index, "reads": int(len(arr)), "ms_per_pass": elapsed_mine / max(1, steps) * 1e3}
    if tim:
        mine.update({"fill_ms": tim["fill_ms"], "fill_launches": int(tim["fill_launches"]), "trace_ms": tim["trace_ms"]})
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    got = table.cpu().numpy().view(np.uint32).copy()  # the last REDUCED table (the barrier drained every stream)
    # the same passes without the collective (each rank's table then stays its own)
    collective_ab = None
    if red and steps > 0:
        env["barrier"]()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_pass(batch, None)
        env["barrier"]()
        plain = env["max_over_ranks"](time.perf_counter() - t0)
        collective_ab = {"ms_per_pass_with": elapsed / steps * 1e3, "ms_per_pass_without": plain / steps * 1e3, "with_vs_without": elapsed / plain}
    n_sites, n_reads = len(sset.sites), int(sset.n_reads_site.sum())
    out = {"config": "configs[2]/[3]: %d mixed DEL / long-DEL / INS sites, 30x paired %dbp reads (%d reads), the same set "
                     "partitioned over %d rank(s) by dist.partition_sites (LPT on reads x graph length); align + count + "
                     "1 all-reduce of the %d-counter table per pass" % (n_sites, sset.L, n_reads, world, n_counters),
           "sites": n_sites, "reads": n_reads, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
           "sites_per_s": n_sites * steps / elapsed, "reads_per_s": n_reads * steps / elapsed, "scaling": "strong",
           "shard_reads": [sset.rank_reads(r) for r in range(world)],
           "shard_imbalance": float(max(sset.rank_weight(r) for r in range(world)) * world / max(1, sset.weights.sum())),
           "hot_sites_split_by_fragment": [{"site": i, "reads": int(sset.n_reads_site[i]),
                                            "reads_per_rank": [len(x) for x in sset.hot_reads[i]]} for i in sset.hot],
           "counters": n_counters, "reduce_equals_single": None, "per_rank": per_rank, "collective_ab": collective_ab,
           # reads/s of two workloads only compare at equal graph length: a read costs 4 x L x G cell updates, and this set's
           # graphs are longer than config 2's 502 columns (insertions up to 1 000 bp, 6-node long deletions)
           "mean_graph_len_per_read": float((sset.n_reads_site * sset.g_len).sum() / max(1, n_reads)),
           "cell_updates_per_s": float(4.0 * sset.L * (sset.n_reads_site * sset.g_len).sum() * steps / elapsed)}
    tall = got[int(graphs.layout.tally_base):].reshape(-1, 4)
    out["tallies"] = {"aligned": int((tall[:, 0] & 0x7FFFFFFF).sum()), "mapped": int(tall[:, 1].sum()),
                      "bad_align": int(tall[:, 2].sum()), "nonuniq": int(tall[:, 3].sum())}
    out["table_sum"] = int(got.astype(np.uint64).sum())
    if world > 1 and rank == 0:
        # 1-rank pass over ALL sites on this rank's device: the reduced table must equal it entry for entry
        arr1, gor1, frag1, rev1 = sset.all_arrays()
        b1 = ctx.new_batch()
        b1.upload(graphs, synth.packed_to_capi(arr1), gor1)
        b1.set_fragments(frag1, rev1)
        t1 = torch.zeros(n_counters, dtype=torch.int32, device=env["device"])
        torch.cuda.synchronize()
        ctx.counts_zero(t1.data_ptr(), n_counters)
        b1.align(capi.AF_ALL)
        b1.count(remove_nonuniq=True, bad_align_frac=0.8, d_counts=t1.data_ptr())
        ctx.sync()
        single = t1.cpu().numpy().view(np.uint32)
        out["reduce_equals_single"] = bool(np.array_equal(single, got))
        out["single_table_sum"] = int(single.astype(np.uint64).sum())
        b1.close()
    if rank == 0 and args.sites_verify > 0 and not args.no_cpu_baseline:
        # a fixed sample of rank 0's whole sites = the first reads of its batch; the table compared is the REDUCED one (the
        # other ranks contribute zeros to these sites' counters)
        sample_idx = [int(i) for i in sset.parts[0][:args.sites_verify]]
        if sample_idx:
            sample = [sset.sites[i] for i in sample_idx]
            res_s, ops_s = batch.download()
            _, sup_s, _ = batch.download_counts(want_table=False)
            info, want = run_sites_cpu_leg(sample)
            out["verified"] = verify_sites(capi, graphs, sample, sample_idx, res_s, ops_s, sup_s, got, want, info)
    batch.close()
    graphs.close()
    return out, tim, sset.b_alg(rank), len(arr)


# ---------------------------------------------------------------------------------------------------
# e2e: BAM files -> genotype documents ("sites genotyped/sec")
# ---------------------------------------------------------------------------------------------------
def prepare_e2e(args, rank, world, ncpu):
    """Writes the e2e data set (paragraph_amd/synth_e2e.py: reference, ONE coordinate-sorted BAM + index, one graph description
    per site, manifest, truth) BEFORE the process touches HIP -- the makers fork.  With N ranks every rank makes the pieces
    of its N-th of the sites and rank 0 joins them; the others wait for the marker file."""
    import pickle
    from paragraph_amd import synth_e2e
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    # one directory per LAUNCH: the ranks of a launch share the rendezvous port and their parent (the torch.distributed.run agent);
    # a directory a crashed earlier launch left behind under the same port is then never mistaken for this one's
    tag = "%s_%d" % (os.environ.get("MASTER_PORT"), os.getppid()) if world > 1 else None
    d = os.path.join(base, "pgbench_e2e_%s" % tag) if tag else tempfile.mkdtemp(prefix="pgbench_e2e_", dir=base)
    os.makedirs(d, exist_ok=True)
    n = args.e2e_sites
    shard0 = list(range(0, n, world))
    stride = max(1, len(shard0) // max(1, args.e2e_verify))
    sample_idx = shard0[::stride][:args.e2e_verify] if args.e2e_verify > 0 and not args.no_cpu_baseline else []
    t0 = time.perf_counter()
    synth_e2e.make_part(d, rank, world, n_sites=n, depth=30.0, seed=1, read_len=args.read_len, procs=max(1, ncpu // world), keep_sites=sample_idx)
    ready = os.path.join(d, ".ready")
    if rank == 0:
        data = synth_e2e.join(d, world, n_sites=n, depth=30.0, seed=1, read_len=args.read_len, wait_s=900.0)
        with open(ready + ".tmp", "wb") as f:
            pickle.dump({"reads": data["reads"], "bam_bytes": data["bam_bytes"]}, f)
        os.replace(ready + ".tmp", ready)
    else:
        deadline = time.time() + 900.0
        while not os.path.exists(ready):
            if time.time() > deadline:
                raise RuntimeError("e2e data set: rank 0 did not finish %s" % d)
            time.sleep(0.05)
        with open(ready, "rb") as f:
            meta = pickle.load(f)
        data = {"reference": os.path.join(d, "ref.fa"), "manifest": os.path.join(d, "manifest.txt"),
                "graphs": [os.path.join(d, "graphs", "site_%d.json" % i) for i in range(n)], "sites": synth_e2e.draw_sites(n, 1),
                "reads": meta["reads"], "bam_bytes": meta["bam_bytes"], "kept": {}, "ref": None}
        data["truth"] = [s.truth() for s in data["sites"]]
    data.update({"dir": d, "sample_idx": sample_idx, "make_s": time.perf_counter() - t0})
    return data


def e2e_reference_sites(e2e, read_len):
    """The sampled sites as the reference's workflow would see them (TEST INFRASTRUCTURE, for reference_site_outcome): the graph
    as GraphInput.cpp loads it, the reads ReadExtraction.cpp keeps, BAM strand flags, fragments by name."""
    from paragraph_amd import synth, synth_e2e
    out = []
    for i in e2e["sample_idx"]:
        spec, reads = e2e["sites"][i], e2e["kept"][i]
        names, seqs, edges, labels = synth_e2e.loaded_graph(spec, e2e["ref"])
        keep = synth_e2e.extracted(spec, reads, read_len)
        site = synth.Site(spec.kind, names, seqs, edges, labels, {})
        out.append(synth.SiteReads(site, np.ascontiguousarray(reads["bases"][keep]), reads["fragment"][keep].astype(np.uint32),
                                   ((reads["flag"][keep] & 0x10) != 0).astype(np.uint8), spec.gt))
    return out


def run_e2e_leg(args, env, e2e, dev_index, shared, ncpu):
    """K timed passes of pgw_genotype_graphs over this rank's sites (r, r + N, ...), barrier + max over ranks around them;
    then -- outside the timed region -- the documents of the last pass are read back: genotypes against the simulated truth
    (all ranks), the per-site edge-count table summed over the ranks (the path's one collective, timed on its own), and on
    rank 0 a sample of sites against the reference's code."""
    import resource
    torch, dist = env["torch"], env["dist"]
    rank, world, device = env["rank"], env["world"], env["device"]
    from paragraph_amd import workflow
    n = len(e2e["graphs"])
    mine = list(range(rank, n, world))
    graphs = [e2e["graphs"][i] for i in mine]
    threads = args.e2e_threads or max(1, ncpu // world)
    out_file = os.path.join(e2e["dir"], "genotypes_rank%d.json" % rank)
    options = {"threads": threads, "devices": [dev_index]}
    if args.e2e_options:
        options.update(json.loads(args.e2e_options))
    if shared:  # ranks sharing a GPU (tests on a 1-GPU box): the host library's workspace budget per rank
        os.environ.setdefault("PG_WORKSPACE_GIB", "%g" % max(4.0, 96.0 / world))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def one_pass():
        workflow.genotype_graphs_to_file(e2e["reference"], e2e["manifest"], graphs, out_file, **options)

    one_pass()  # warm-up: the host library's own device context, its workspace, the file cache
    barrier()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    th0 = _cpu_throttle()
    t0 = time.perf_counter()
    for _ in range(args.e2e_steps):
        one_pass()
    barrier()
    elapsed_mine = time.perf_counter() - t0
    th1 = _cpu_throttle()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    elapsed = env["max_over_ranks"](elapsed_mine)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)

    with open(out_file) as f:
        docs = json.load(f)
    os.unlink(out_file)
    assert len(docs) == len(mine)
    # The same job with `paragraph`'s default cascade (src/c++/main/paragraph.cpp:60-61: exact path matching first, gssw on what it
    # leaves): path stage -> filter chain -> hand-over on the device -> gssw stage, composed inside the workflow.  As many passes (two at least).
    cascade = None
    if args.e2e_steps > 0:
        options_path = dict(options, path_sequence_matching=True)
        workflow.genotype_graphs_to_file(e2e["reference"], e2e["manifest"], graphs, out_file, **options_path)
        barrier()
        rp0 = resource.getrusage(resource.RUSAGE_SELF)
        thp0 = _cpu_throttle()
        t0 = time.perf_counter()
        path_steps = max(2, args.e2e_steps)
        for _ in range(path_steps):
            workflow.genotype_graphs_to_file(e2e["reference"], e2e["manifest"], graphs, out_file, **options_path)
        barrier()
        t_path = env["max_over_ranks"](time.perf_counter() - t0)
        rp1 = resource.getrusage(resource.RUSAGE_SELF)
        thp1 = _cpu_throttle()
        cpu_path = (rp1.ru_utime - rp0.ru_utime) + (rp1.ru_stime - rp0.ru_stime)
        with open(out_file) as f:
            docs_path = json.load(f)
        os.unlink(out_file)
        same_gt = sum(1 for a, b2 in zip(docs, docs_path) if a["samples"]["SYN"]["gt"].get("GT") == b2["samples"]["SYN"]["gt"].get("GT"))
        cascade = {"sites_genotyped_per_s": n * path_steps / t_path, "ms_per_step": t_path / path_steps * 1e3, "steps": path_steps,
                   "cpu_us_per_site_sample_this_rank": cpu_path / path_steps / max(1, len(mine)) * 1e6,
                   "genotypes_equal_the_gssw_only_run_on_this_rank": same_gt, "sites_on_this_rank": len(mine),
                   "cpu_throttling_this_rank": _throttle_delta(thp0, thp1),
                   "note": "path_sequence_matching = true (the `paragraph` tool's default): reads the exact path matcher maps and the "
                           "filters accept keep that alignment, so counts may differ from the gssw-only run by design"}
    # grmpy's own cascade (gssw only) with the exact shortcut in front of it (BatchParameters::exact_match_shortcut,
    # pg_batch_retire_exact_matches): the documents must be the gssw-only run's, byte for byte of their JSON values
    shortcut = None
    if args.e2e_steps > 0 and not args.no_e2e_shortcut:
        options_sc = dict(options, exact_match_shortcut=True)
        workflow.genotype_graphs_to_file(e2e["reference"], e2e["manifest"], graphs, out_file, **options_sc)
        barrier()
        rs0 = resource.getrusage(resource.RUSAGE_SELF)
        ths0 = _cpu_throttle()
        t0 = time.perf_counter()
        sc_steps = max(2, args.e2e_steps)
        for _ in range(sc_steps):
            workflow.genotype_graphs_to_file(e2e["reference"], e2e["manifest"], graphs, out_file, **options_sc)
        barrier()
        t_sc = env["max_over_ranks"](time.perf_counter() - t0)
        rs1 = resource.getrusage(resource.RUSAGE_SELF)
        ths1 = _cpu_throttle()
        with open(out_file) as f:
            docs_sc = json.load(f)
        os.unlink(out_file)
        shortcut = {"sites_genotyped_per_s": n * sc_steps / t_sc, "ms_per_step": t_sc / sc_steps * 1e3, "steps": sc_steps,
                    "cpu_us_per_site_sample_this_rank": ((rs1.ru_utime - rs0.ru_utime) + (rs1.ru_stime - rs0.ru_stime)) / sc_steps / max(1, len(mine)) * 1e6,
                    "documents_equal_the_gssw_only_run_on_this_rank": sum(1 for a, b2 in zip(docs, docs_sc) if a == b2),
                    "sites_on_this_rank": len(mine), "cpu_throttling_this_rank": _throttle_delta(ths0, ths1),
                    "note": "exact_match_shortcut = true: reads whose alignRead record one exact full-length match forces skip their "
                            "four fills (pg_batch_retire_exact_matches); every document -- counts, statistics, genotypes -- must equal "
                            "the gssw-only run's"}
    # ... and with all four stages of the cascade on (path -> k-mer -> klib -> gssw; `paragraph --kmer-sequence-matching
    # --klib-sequence-matching`, both default OFF in the reference's tools): two passes, reported only -- a failure here is written
    # into the line, it does not fail the run
    all_four = None
    if args.e2e_steps > 0 and world == 1:  # (N = 1 only: a rank that failed here alone would leave the others at a barrier)
        try:
            options_all = dict(options, path_sequence_matching=True, kmer_sequence_matching=True, klib_sequence_matching=True)
            workflow.genotype_graphs_to_file(e2e["reference"], e2e["manifest"], graphs, out_file, **options_all)
            barrier()
            t0 = time.perf_counter()
            for _ in range(2):
                workflow.genotype_graphs_to_file(e2e["reference"], e2e["manifest"], graphs, out_file, **options_all)
            barrier()
            t_all = env["max_over_ranks"](time.perf_counter() - t0)
            with open(out_file) as f:
                docs_all = json.load(f)
            os.unlink(out_file)
            all_four = {"sites_genotyped_per_s": n * 2 / t_all, "ms_per_step": t_all / 2 * 1e3, "steps": 2,
                        "genotypes_equal_the_gssw_only_run_on_this_rank": sum(
                            1 for a, b2 in zip(docs, docs_all) if a["samples"]["SYN"]["gt"].get("GT") == b2["samples"]["SYN"]["gt"].get("GT")),
                        "sites_on_this_rank": len(mine)}
        except Exception as exc:  # noqa: BLE001 -- reported, not fatal
            all_four = {"error": "%s: %s" % (type(exc).__name__, exc)}
    concordant = sum(1 for i, doc in zip(mine, docs) if doc["samples"]["SYN"]["gt"].get("GT") == e2e["truth"][i]["gt"])
    errors = sum(1 for doc in docs if "error" in doc)
    # per-site edge-count table in one layout on every rank: a slot per edge of every site's graph, in site order
    edge_names, offsets, total = [], [], 0
    for spec in e2e["sites"]:
        g = spec.graph()
        edge_names.append(["%s_%s" % (e["from"], e["to"]) for e in g["edges"]])
        offsets.append(total)
        total += len(g["edges"])
    table = np.zeros(total + 2, dtype=np.int32)  # + (concordant genotypes, documents with an error)
    for i, doc in zip(mine, docs):
        counts = {}
        for bp in doc["samples"]["SYN"]["breakpoints"].values():
            counts.update(bp["counts"]["edges"])
        for k, name in enumerate(edge_names[i]):
            table[offsets[i] + k] = counts.get(name, 0)
    table[total], table[total + 1] = concordant, errors
    own = table.copy()
    reduce_ms = None
    if world > 1 or env["reducer"] is not None:
        t = torch.from_numpy(table)
        if not shared:
            t = t.to(device)
        barrier()
        t0 = time.perf_counter()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        barrier()
        reduce_ms = (time.perf_counter() - t0) * 1e3
        table = t.cpu().numpy()
    mine_kept = all(np.array_equal(table[offsets[i]:offsets[i] + len(edge_names[i])], own[offsets[i]:offsets[i] + len(edge_names[i])]) for i in mine)
    per_rank = [{"rank": rank, "sites": len(mine), "seconds": elapsed_mine, "cpu_s": cpu_s, "threads": threads}]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "sites": len(mine), "seconds": elapsed_mine, "cpu_s": cpu_s, "threads": threads})
    if rank != 0:
        return None
    steps = args.e2e_steps
    cpu_all = sum(p["cpu_s"] for p in per_rank)
    out = {"config": "BAM -> genotypes: %d synthetic del / ins / swap sites (graph descriptions with reference-interval nodes), one 30x "
                     "sample of paired %dbp reads in ONE coordinate-sorted BAM (%d reads, %.0f MB), pgw_genotype_graphs = BGZF + BAM "
                     "decode, read extraction (ReadExtraction.cpp policy), graph loading, device batches (4 fills + traceback + "
                     "filters + counts per read), count documents, breakpoint genotyping, the JSON array written to a file; "
                     "rank r takes sites r, r + %d, ..." % (n, args.read_len, e2e["reads"], e2e["bam_bytes"] / 1e6, world),
           "sites": n, "samples": 1, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
           "sites_genotyped_per_s": n * steps / elapsed, "reads_in_bam_per_s": e2e["reads"] * steps / elapsed, "scaling": "strong",
           "host_threads": threads * world, "host_threads_per_rank": threads, "cpu_quota_cores": _cpu_quota(), "host_cpus": ncpu,
           "cpu_s_per_pass": cpu_all / steps, "cpu_us_per_site_sample": cpu_all / steps / n * 1e6,
           "cpu_throttling_rank0": _throttle_delta(th0, th1),
           "cpu_note": "user + system time of the rank processes over the timed passes (getrusage): every host thread of the workflow, "
                       "the HIP runtime's threads included",
           "genotypes_equal_truth": int(table[total]), "documents_with_error": int(table[total + 1]),
           "edge_table": {"entries": total, "sum": int(table[:total].astype(np.int64).sum()), "reduce_ms": reduce_ms,
                          "reduced_equals_own_on_own_sites": bool(mine_kept),
                          "note": "one slot per edge of every site (fragment counts of the breakpoint edges), all-reduced over the ranks "
                                  "AFTER the timed passes: a site's genotype needs only its own counts, the sum only collects them"},
           "data_make_s": e2e["make_s"], "per_rank": per_rank, "with_path_matching": cascade, "with_exact_shortcut": shortcut,
           "with_all_four_stages": all_four}
    # what the host allows: with C usable cores and c CPU-seconds per (site, sample) no more than C / c sites per second leave the
    # node however many devices it has -- the first thing to read off an N-GPU curve of this leg
    cores = float(out["cpu_quota_cores"] or ncpu)
    cpu_per_site = out["cpu_us_per_site_sample"] * 1e-6
    out["host_bound"] = {"usable_cores": cores, "ceiling_sites_per_s": cores / cpu_per_site if cpu_per_site > 0 else None,
                         "cores_busy": out["sites_genotyped_per_s"] * cpu_per_site,
                         "cores_per_rank_at_this_rate": out["sites_genotyped_per_s"] * cpu_per_site / max(1, world),
                         "threads_per_rank": threads,
                         "note": "ceiling = usable cores / CPU seconds per (site, sample); a measured rate near it is the host's, not the "
                                 "devices'"}
    # A genotype that differs from the simulated truth is not by itself an error of the path (30x sampling can starve an allele);
    # a concordance below 99.5 % is.  What must hold exactly: no document with an error, the table checks, the sampled sites.
    out["genotype_concordance"] = int(table[total]) / max(1, n)
    bad = (1 if out["genotype_concordance"] < 0.995 else 0) + int(table[total + 1]) + (0 if mine_kept else 1)
    if e2e["sample_idx"]:
        sample = e2e_reference_sites(e2e, args.read_len)
        info, want = run_sites_cpu_leg(sample)
        wrong, first = 0, None
        for i, s, w in zip(e2e["sample_idx"], sample, want):
            ref_counts = {"%s_%s" % (s.site.names[a], s.site.names[b]): int(w["edge_counts"][k][0]) for k, (a, b) in enumerate(s.site.edges)}
            got = {name: int(table[offsets[i] + k]) for k, name in enumerate(edge_names[i])}
            # the genotype document reports the edges of its breakpoints (every edge but source -> LF and RF -> sink here)
            diff = {k: (got[k], ref_counts[k]) for k in got if not k.startswith("source_") and not k.endswith("_sink") and got[k] != ref_counts[k]}
            if diff:
                wrong += 1
                if first is None:
                    first = {"site": i, "edges (got, reference)": diff}
        out["verified"] = {"sites": len(sample), "reads": int(sum(len(s.reads) for s in sample)), "site_mismatches": wrong, "first_bad_site": first,
                           "what": "fragment count of every breakpoint edge of the sampled sites (from the reduced table) = the reference's "
                                   "gssw.c alignments + graph-tools counting on the reads ReadExtraction.cpp:135-178 keeps",
                           "reference": info}
        bad += wrong
    out["mismatches"] = int(bad)
    return out


def cleanup_e2e(e2e, rank, world, dist):
    import shutil
    if world > 1:
        dist.barrier()
    if rank == 0:
        shutil.rmtree(e2e["dir"], ignore_errors=True)


# ---------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------
_JSON_FD = None


def quiet_stdout():
    """Everything but the result line goes to stderr: native libraries print to file descriptor 1 behind Python's back (RCCL
    writes its version banner there when a communicator is created, flushed at exit -- i.e. AFTER the JSON line).  fd 1 is
    pointed at stderr for the life of the process; the JSON line is written to the saved original."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def print_result_line(obj):
    data = (json.dumps(obj) + "\n").encode()
    fd = _JSON_FD if _JSON_FD is not None else 1
    while data:
        data = data[os.write(fd, data):]


def main_rank(args):
    quiet_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from paragraph_amd import synth
    L = args.read_len
    headline3 = args.workload == "config3"
    want_sites = headline3 or args.sites_steps > 0

    # ---- data first: the generators fork workers, which must happen before this process touches HIP ------------
    ncpu = _effective_cpus()  # affinity cut to the cgroup's quota, shared by the ranks of this node
    site = arr = None
    if not headline3:
        log("generating config2 reads")
        site, arr = synth.config2_reads_packed(args.reads, read_len=L, seed=2 + rank)
    sset = None
    if want_sites:
        log("generating the config3 site set")
        sites = synth.mixed_sites_parallel(args.sites, seed=3, procs=max(1, min(16, ncpu // max(1, world))), read_len=L)
        if args.hot_site_depth > 0:  # one deep site on top of the set: split over the ranks by fragment id
            sites = sites + synth.mixed_sites(1, seed=11, read_len=L, depth=args.hot_site_depth, site_streams=True)
        sset = SiteSet(sites, L, world, split_reads=args.split_reads)
    e2e = None
    if args.e2e_steps > 0:
        log("writing the e2e data set")
        e2e = prepare_e2e(args, rank, world, ncpu)
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
    backend, pg_error = None, None
    want_pg = world > 1 or args.collective in ("auto", "on")
    if want_pg:
        backend = "gloo" if shared else "nccl"
        if world == 1:  # a communicator of one rank: the RCCL code path of N = 8 on the one GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        try:
            if shared:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=device)
        except Exception as e:  # noqa: BLE001
            if world > 1 or args.collective == "on":
                raise
            backend, pg_error = None, "%s: %s" % (type(e).__name__, e)
        if backend:
            log("process group: backend %s, world %d, device %d%s" % (dist.get_backend(), dist.get_world_size(), dev_index, " (shared)" if shared else ""))

    from paragraph_amd import capi
    from paragraph_amd import dist as pgdist
    ws_gib = args.workspace_gib
    if shared:
        ws_gib = min(ws_gib, max(8.0, 160.0 / ((world + ndev - 1) // ndev)))
    ctx = capi.Context(dev_index, workspace_bytes=int(ws_gib * (1 << 30)))
    reducer = pgdist.CountReduce(ctx, device) if backend else None

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
           "max_over_ranks": max_over_ranks, "reducer": reducer}
    dist_info = {"world": world, "backend": backend, "shared_device": bool(shared), "devices_visible": ndev,
                 "launcher": os.environ.get("PG_BENCH_LAUNCHER", "external torch.distributed.run" if world > 1 else "single process"),
                 "collective_in_step": bool(reducer), "collective": None if not reducer else
                 ("all_reduce(SUM) of the int32 counter table on a stream of its own, ordered by events against the library's "
                  "count stream (paragraph_amd.dist.CountReduce): no host synchronisation inside a step, two tables take turns"
                  if not reducer.blocking else "host hop under gloo (ranks share a device): compute streams drained, then reduced"),
                 "process_group_error": pg_error}
    if backend:
        dist_info["backend"] = dist.get_backend()
        dist_info["world"] = dist.get_world_size()
        prop = torch.cuda.get_device_properties(dev_index)
        mine = {"rank": rank, "local_rank": local_rank, "device": dev_index, "name": torch.cuda.get_device_name(dev_index),
                "host": socket.gethostname(), "pci": "%s:%s:%s" % tuple(getattr(prop, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")),
                "uuid": str(getattr(prop, "uuid", "")) or None}
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, mine)
            dist_info["ranks"] = every
        else:
            dist_info["ranks"] = [mine]
        # One process per GPU means just that: with at least as many GPUs as ranks every rank must sit on a device of its own and
        # the reduce must be RCCL's.  A launch that does not (a wrong LOCAL_RANK, a masked device list) is reported and fails.
        ids = [(r["host"], r["device"]) for r in dist_info["ranks"]]
        hw = [(r["host"], r["uuid"] or r["pci"]) for r in dist_info["ranks"]]
        dist_info["distinct_devices"] = len(set(ids)) == len(ids) and (len(set(hw)) == len(hw) or all(h[1] in (None, "None:None:None") for h in hw))
        dist_info["placement_ok"] = bool(shared or (dist_info["distinct_devices"] and dist_info["backend"] == "nccl"))

    out = None
    if headline3:
        sites_out, tim, b_alg_mine, reads_mine = run_sites_leg(args, env, ctx, capi, synth, sset, args.steps, args.warmup, True)
        if rank == 0:
            fill_s = tim["fill_ms"] / 1e3
            achieved = b_alg_mine * args.steps / fill_s / 1e9 if fill_s > 0 else 0.0
            out = {
                "metric": "150bp reads aligned/sec (whole node)", "value": sites_out["reads_per_s"], "unit": "reads/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sites_out["ms_per_step"],
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE,
                "data": "synthetic",
                "config": {"workload": sites_out["config"], "sites": sites_out["sites"], "reads": sites_out["reads"],
                           "read_len": L, "graph_len": float(np.mean(sset.g_len)), "parallelism": "sites x%d" % world,
                           "sites_per_s": sites_out["sites_per_s"]},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                             "kernel": "pg_fill_kernel<%d, false, 16>" % (2 * ((L + 31) // 32)),
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
        if args.config5_graphs > 0:
            log("config5 leg")
            c5 = run_config5_leg(args, env, ctx, capi)
            if rank == 0:
                out["config5"] = c5
    if e2e is not None:
        log("e2e leg")
        barrier()
        ctx.close()  # the workflow's host library opens its own context (own workspace budget) on the same device
        e2e_out = run_e2e_leg(args, env, e2e, dev_index, shared, ncpu)
        cleanup_e2e(e2e, rank, world, dist)
        if rank == 0:
            out["e2e"] = e2e_out
    rc = 0
    if rank == 0:
        print_result_line(out)
        if out.get("verified") and out["verified"]["mismatches"]:
            rc = 3
        if out.get("e2e") and out["e2e"]["mismatches"]:
            rc = 3
        pl = out.get("plain_stage")
        if pl and (pl["records_differing_from_the_lean_step"] or not pl["cigar_strings_equal"]):
            rc = 3
        sc = out.get("exact_shortcut")
        if sc and (sc["records_differing_from_the_plain_step"] or not sc["cigar_elements_equal"] or not sc["count_table_equal"]):
            rc = 3
        if out.get("config5") and out["config5"].get("verified") and out["config5"]["verified"]["mismatches"]:
            rc = 3
        if out.get("sites") and out["sites"].get("reduce_equals_single") is False:
            rc = 3
        if out.get("sites") and out["sites"].get("verified") and out["sites"]["verified"]["mismatches"]:
            rc = 3
        if dist_info.get("placement_ok") is False:
            print("bench.py: %d ranks on %d visible GPUs must each have a device of their own under nccl: %s"
                  % (world, ndev, json.dumps(dist_info.get("ranks"))), file=sys.stderr)
            rc = 4
    if backend:
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

    # per-site counter table {count, READS, FWD, REV} x (nodes, edges, sequence sets) + filter tallies: torch tensors so that
    # the reduce is ONE RCCL all-reduce over xGMI on device memory, no host hop.  Two tables take turns like the batches:
    # step n + 1 zeroes and fills the other one while reduce n runs.
    n_counters = int(graphs.layout.n_counters)
    tables = [torch.zeros(n_counters, dtype=torch.int32, device=device) for _ in range(2)]
    torch.cuda.synchronize()
    red = env["reducer"]
    step_no = [0]

    def step(red):
        k = step_no[0] & 1
        b, t = batches[k], tables[k]
        step_no[0] += 1
        if red:
            red.acquire(t)  # the count stream waits for the event behind this table's previous reduce; the host does not
        ctx.counts_zero(t.data_ptr(), n_counters)  # on the stream the count kernels run on: ordered with them
        b.align(capi.AF_ALL)
        b.count(remove_nonuniq=True, bad_align_frac=0.8, d_counts=t.data_ptr())
        if red:
            red.reduce(t)  # behind an event of the count stream, on a stream of its own: nothing blocks the host here

    def timed(red, steps):
        env["barrier"]()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(red)
        env["barrier"]()
        return env["max_over_ranks"](time.perf_counter() - t0)

    for _ in range(args.warmup):
        step(red)
    env["barrier"]()
    log("warmup done")
    ctx.timing_enable(True)
    ctx.timing_reset()
    elapsed = timed(red, args.steps)
    log("timed region %.3fs" % elapsed)
    tim = ctx.timing()
    ctx.timing_enable(False)
    counts_t = tables[(step_no[0] - 1) & 1]
    tab = counts_t.cpu().numpy().view(np.uint32).copy()
    last_batch = batches[(step_no[0] - 1) & 1]
    # results of the last timed step (PCIe-inclusive figure, verification)
    batch = last_batch
    t0 = time.perf_counter()
    res, ops = batch.download()
    t_download = time.perf_counter() - t0
    site_counts = capi.decode_counts(graphs, tab)[0]

    # A/B for the collective: the same steps again without it (N = 1 only: there it must cost nothing)
    collective = None
    if red:
        elapsed_plain = timed(None, args.steps)
        collective = {"reduces_in_timed_region": args.steps, "ms_per_step_with": elapsed / args.steps * 1e3,
                      "ms_per_step_without": elapsed_plain / args.steps * 1e3, "with_vs_without": elapsed / elapsed_plain,
                      "note": "`value` is measured WITH the all-reduce in every step (at N = 1 through a world-size-1 RCCL "
                              "communicator = the code path of N = 8); `without` = the same steps right after, no collective"}
        log("plain region %.3fs" % elapsed_plain)
    # Reported only: the same steps through the PLAIN gssw stage (pg_ctx_set_lean(0): four fills per read, all four multi flags), and
    # the lean stage's records of the last timed step against its records -- every field of pg_result; of multi_mask the bits of the
    # fills that ran
    plain_stage = None
    if args.plain_steps > 0 and (tim.get("lean_rev_launches", 0) > 0 or tim.get("lean_fused_launches", 0) > 0):
        ctx.set_lean(False)
        for _ in range(min(1, args.warmup)):
            step(red)
        env["barrier"]()
        ctx.timing_enable(True)
        ctx.timing_reset()
        elapsed_pl = timed(red, args.plain_steps)
        tim_pl = ctx.timing()
        ctx.timing_enable(False)
        res_pl, ops_pl = batches[(step_no[0] - 1) & 1].download()
        ctx.set_lean(True)
        # what bounds the plain stage's fill kernel, from counter files of THIS build's kernel sources collected with PG_LEAN=0
        import copy
        args_pl = copy.copy(args)
        args_pl.traffic_json, args_pl.sq_json = args.traffic_json_plain, args.sq_json_plain
        fill_pl_s = tim_pl["fill_ms"] / 1e3
        launches_pl = max(1, tim_pl["fill_launches"])
        bounds_pl = measured_bounds(args_pl, args.reads * args.plain_steps / launches_pl, fill_pl_s / launches_pl)
        differ = np.zeros(len(res), dtype=bool)
        for f in ("graph_pos", "score", "mapq", "is_unique", "returned_reverse", "n_ops", "clipped", "status"):
            differ |= res[f] != res_pl[f]
        differ |= (res["strand_score"] != res_pl["strand_score"]).any(axis=1)
        skipped = (res["multi_mask"] & 0x10) != 0
        other_bit = np.where(res["returned_reverse"] != 0, 1, 2).astype(np.uint8)  # bit of the forward fill of the strand not returned
        mask = np.where(skipped, 0x0F & ~other_bit, 0x0F).astype(np.uint8)
        differ |= (res["multi_mask"] & mask) != (res_pl["multi_mask"] & mask)
        def flat(r, o):  # every read's CIGAR elements, read after read (ops_off is an allocation order, not content)
            n_ops = r["n_ops"].astype(np.int64)
            first = np.cumsum(n_ops) - n_ops
            return o[np.repeat(r["ops_off"].astype(np.int64) - first, n_ops) + np.arange(int(n_ops.sum()), dtype=np.int64)]

        cig_same = bool(np.array_equal(flat(res, ops), flat(res_pl, ops_pl))) if not differ.any() else False
        plain_stage = {"reads_per_s": args.reads * world * args.plain_steps / elapsed_pl, "ms_per_step": elapsed_pl / args.plain_steps * 1e3,
                       "steps": args.plain_steps, "value_vs_plain": (args.reads * world * args.steps / elapsed) / (args.reads * world * args.plain_steps / elapsed_pl),
                       "records_differing_from_the_lean_step": int(differ.sum()), "cigar_strings_equal": cig_same,  # (their elements, read after read)
                       "forward_fills_of_the_other_strand_not_run": int(skipped.sum()),
                       "kernel": "pg_fill_kernel<%d, false, 16>" % (2 * ((L + 31) // 32)), "launches": int(tim_pl["fill_launches"]),
                       "avg_launch_ms": tim_pl["fill_ms"] / launches_pl,
                       "hbm_measured_frac": bounds_pl.get("hbm_measured_frac"), "hbm_measured_gbs": bounds_pl.get("hbm_measured_gbs"),
                       "traffic_source": bounds_pl.get("traffic_source"),
                       "valu_issue_frac_with_measured_pairing": (bounds_pl.get("valu") or {}).get("issue_frac_with_measured_pairing"),
                       "hbm_alg_h_only_frac": (args.reads * args.plain_steps * (2 * L * G + L + 64) / fill_pl_s / 1e9 / HBM_PEAK_GBS) if fill_pl_s > 0 else None,
                       "note": "pg_ctx_set_lean(0): GraphAligner::alignRead's four fills for every read; same reads, same batches, right "
                               "after the timed region.  Compared: every pg_result field (ops_off aside) and the rendered CIGAR strings"}
        log("plain stage %.3fs" % elapsed_pl)
    # Reported only, never `value`: the same workload with the EXACT shortcut in front of the gssw stage
    # (pg_batch_retire_exact_matches, include/paragraph_amd.h).  The path kernel runs over every read; a read whose
    # alignRead(AF_ALL) record its one exact full-length match forces keeps that record and skips its four fills, every other
    # read is aligned as in the plain step.  Compared with the plain step's output on every read: all fields of the reference's
    # Read (position, score, MAPQ, uniqueness, strand, CIGAR elements) and the whole count table.
    shortcut = None
    if args.exact_shortcut_steps > 0 and L <= 250 and world == 1 and not args.shortcut_in_process:
        # In a process of its own: the leg's path stages make the context's second seed stream, and what that leaves with the HIP runtime
        # of THIS process outlives the context -- the e2e leg behind it then read 72 k instead of 78 k sites/s, its all-four-stages pass
        # 22 k instead of 30 k (profiles/r06_shortcut_leg_in_process_ab.jsonl).  The child runs the same batch (same seed) through the
        # lean stage once, then the leg, and compares the two itself.
        cmd = [sys.executable, os.path.abspath(__file__), "--shortcut-in-process", "--reads", str(args.reads), "--read-len", str(L), "--steps", "1",
               "--warmup", "1", "--plain-steps", "0", "--no-cpu-baseline", "--sites-steps", "0", "--config5-graphs", "0", "--e2e-steps", "0",
               "--stream-batches", "0", "--collective", "off", "--exact-shortcut-steps", str(args.exact_shortcut_steps), "--workspace-gib", "64"]
        cenv = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            cenv.pop(k, None)
        try:
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=cenv, timeout=600)
            child = json.loads([l for l in pr.stdout.decode().splitlines() if l.startswith("{")][-1])
            shortcut = child.get("exact_shortcut")
            if shortcut:
                shortcut["in_a_process_of_its_own"] = True
                shortcut["lean_step_of_that_process_reads_per_s"] = child.get("value")
        except Exception as e:  # noqa: BLE001
            shortcut = None
            log("exact_shortcut child failed: %s" % e)
    if args.exact_shortcut_steps > 0 and L <= 250 and world == 1 and args.shortcut_in_process:  # (N = 1 only, like the streaming leg: `tab` is the reduced table at N > 1)
        t0 = time.perf_counter()
        graphs.build_path_index(32)
        t_index = time.perf_counter() - t0

        def step_shortcut():
            k = step_no[0] & 1
            b, t = batches[k], tables[k]
            step_no[0] += 1
            ctx.counts_zero(t.data_ptr(), n_counters)
            b.set_active(None)
            b.path_align(fetch_flags=False)
            b.retire_exact_matches()
            b.align(capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS)
            b.count(remove_nonuniq=True, bad_align_frac=0.8, d_counts=t.data_ptr())

        for _ in range(2):
            step_shortcut()
        env["barrier"]()
        ctx.timing_enable(True)
        ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(args.exact_shortcut_steps):
            step_shortcut()
        env["barrier"]()
        t_sc = env["max_over_ranks"](time.perf_counter() - t0)
        tim_sc = ctx.timing()
        ctx.timing_enable(False)
        res2, ops2 = batches[(step_no[0] - 1) & 1].download()
        tab2 = tables[(step_no[0] - 1) & 1].cpu().numpy().view(np.uint32)

        def flat_ops(r, o):  # every read's CIGAR elements, read after read (ops_off is an allocation order, not content)
            n_ops = r["n_ops"].astype(np.int64)
            first = np.cumsum(n_ops) - n_ops
            idx = np.repeat(r["ops_off"].astype(np.int64) - first, n_ops) + np.arange(int(n_ops.sum()), dtype=np.int64)
            return o[idx]

        skipped = (res2["strand_score"] == -1).any(axis=1)
        fields = ("graph_pos", "score", "mapq", "is_unique", "returned_reverse", "n_ops", "clipped", "status")
        bad = np.zeros(len(res), dtype=bool)
        for f in fields:
            bad |= res[f] != res2[f]
        # (multi_mask: the flags of the fills BOTH runs made -- the lean stage runs the forward-graph fill of the strand it does not return
        #  only where the record needs it, and for a run's last pair without a partner, which depends on which reads are still active)
        either_skipped = ((res["multi_mask"] | res2["multi_mask"]) & 0x10) != 0
        other_bit = np.where(res["returned_reverse"] != 0, 1, 2).astype(np.uint8)
        mm_mask = np.where(either_skipped, 0x0F & ~other_bit, 0x0F).astype(np.uint8)
        bad |= ~skipped & (((res["multi_mask"] & mm_mask) != (res2["multi_mask"] & mm_mask)) | (res["strand_score"] != res2["strand_score"]).any(axis=1))
        same_cigars = bool(np.array_equal(flat_ops(res, ops), flat_ops(res2, ops2))) if not bad.any() else False
        shortcut = {"reads_per_s": args.reads * world * args.exact_shortcut_steps / t_sc, "ms_per_step": t_sc / args.exact_shortcut_steps * 1e3,
                    "steps": args.exact_shortcut_steps, "vs_value": (args.reads * world * args.exact_shortcut_steps / t_sc) / (args.reads * world * args.steps / elapsed),
                    "reads_that_skipped_their_fills": int(skipped.sum()), "skipped_frac": float(skipped.mean()),
                    "fill_ms_per_step": tim_sc["fill_ms"] / args.exact_shortcut_steps, "path_index_build_s": t_index,
                    "records_differing_from_the_plain_step": int(bad.sum()), "cigar_elements_equal": same_cigars,
                    "count_table_equal": bool(np.array_equal(tab, tab2)),
                    "note": "reported only (`value` is the plain step: four fills for every read).  Synthetic reads carry 1 % substitutions per base: "
                            "0.99^150 = 22 % are exact; the index build (host) is outside the timed steps, as the graph upload is"}
        log("exact shortcut leg: %.3fs, %d reads skipped their fills, %d records differ" % (t_sc, int(skipped.sum()), int(bad.sum())))
    # every rank's own view of the timed region: a straggler GPU shows here, not in the max
    mine = {"rank": rank, "device": env["device"].index, "fill_ms_per_launch": tim["fill_ms"] / max(1, tim["fill_launches"]),
            "fill_launches": int(tim["fill_launches"]), "fill_ms": tim["fill_ms"], "trace_ms": tim["trace_ms"]}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

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
    b_alg_h = 2 * L * G + L + 64      # the same with H only (the kernel re-derives E / F in the traceback)
    b_alg_h_lean = L * G + L + 64     # ... and with ONE traced fill per read, which is all the lean stage's record needs
    # (lean stage: a chunk's forward launch runs on a stream of its own beside the next chunk's reversed-graph launch, so the sum of
    #  the launches' durations can exceed the wall clock; the fraction of peak is then taken against the longer of the two: conservative)
    lean_fused = tim.get("lean_fused_launches", 0) > 0
    lean_on = tim.get("lean_rev_launches", 0) > 0 or lean_fused
    fill_s = min(tim["fill_ms"] / 1e3, elapsed) if lean_on and not lean_fused else tim["fill_ms"] / 1e3
    reads_per_fill_leg = args.reads * args.steps  # this rank's fill launches
    achieved_gbs = reads_per_fill_leg * b_alg / fill_s / 1e9 if fill_s > 0 else 0.0
    launches = max(1, tim["fill_launches"])
    avg_launch_s = fill_s / launches
    roof_extra = measured_bounds(args, reads_per_fill_leg / launches, avg_launch_s)
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
        "dtype": DTYPE,
        "data": "synthetic",
        # the PCIe-inclusive rate (pinned host arrays in, every result array back on the host, double-buffered): never `value`
        "value_streaming": (args.reads / t_stream) if t_stream else None,
        "config": {
            "workload": "configs[1]: 1 DEL graph (200bp flanks, nodes 201/100/201), %d synthetic %dbp reads per GPU, "
                        "GraphAligner::alignRead(AF_ALL) per read -- the lean gssw stage: the record the reference reads off its 4 fills from the 3 "
                        "(4 where the better strand is not unique and the other may be: under 0.1 %% of these reads) that can change it, strand pick + traceback -- then "
                        "filters + node/edge/sequence counts%s"
                        % (args.reads, L, " + 1 all-reduce of the counter table per step" if world > 1 else ""),
            "reads_per_gpu": args.reads, "read_len": L, "graph_len": G, "parallelism": "reads x%d" % world,
        },
        # whole step (fill + traceback + count), all ranks: the reference's four fills per read (what the records stand for) ...
        "cell_updates_per_s": 4.0 * L * G * reads_total / elapsed,
        # ... and the cells the device did update (lean stage: three fills per read; the fourth fills of a few per cent not counted)
        "cell_updates_computed_per_s": tim["cells"] * world / elapsed,
        "roofline": {
            **roofline_head(roof_extra, achieved_gbs),
            "kernel": ("pg_fill_lean_fused_kernel<%d> (the lean gssw stage in one launch: per wavefront the reversed-graph fills of two work-item "
                       "pairs, the pick, and the forward-graph fills of their eight higher-scoring strands)" % (2 * ((L + 31) // 32))) if lean_fused
                      else ("pg_fill_lean_kernel<%d, 2> (reversed-graph fills of both strands) + pg_fill_lean_kernel<%d, 3> (forward-graph fills of "
                            "the instance items): the lean gssw stage's two fill launches per chunk, taken together" % ((2 * ((L + 31) // 32),) * 2))
                      if lean_on else "pg_fill_kernel<%d, false, 16>" % (2 * ((L + 31) // 32)),
            "launches": int(tim["fill_launches"]),
            "lean": {"rev_launches": int(tim["lean_rev_launches"]), "rev_ms": tim["lean_rev_ms"], "fwd_launches": int(tim["lean_fwd_launches"]),
                     "fwd_ms": tim["lean_fwd_ms"],
                     "note": "durations by HIP events on each launch's own stream; the forward launch of a chunk overlaps the reversed-graph "
                             "launch of the next one in time"} if lean_on and not lean_fused else None,
            "avg_launch_ms": tim["fill_ms"] / max(1, tim["fill_launches"]),
            **roof_extra,
            "alg_bytes_per_read": b_alg,
            "alg_bytes_per_launch": b_alg * reads_per_fill_leg / max(1, tim["fill_launches"]),
            # what an H-only trace has to move: one byte of H per cell of the two traced fills + the read in + the record out
            "hbm_alg_h_only_bytes_per_read": b_alg_h,
            "hbm_alg_h_only_gbs": reads_per_fill_leg * b_alg_h / fill_s / 1e9 if fill_s > 0 else 0.0,
            "hbm_alg_h_only_frac": (reads_per_fill_leg * b_alg_h / fill_s / 1e9 / HBM_PEAK_GBS) if fill_s > 0 else 0.0,
            # the lean stage traces ONE fill per read (the fourth fills of a few per cent aside): what it has to move is half of that --
            # it is faster BECAUSE it moves fewer bytes, and its share of the HBM peak is lower for the same reason (plain_stage
            # carries the four-fill stage's figures of the same run)
            "hbm_alg_h_only_one_traced_fill_bytes_per_read": b_alg_h_lean if lean_on else None,
            "hbm_alg_h_only_one_traced_fill_frac": (reads_per_fill_leg * b_alg_h_lean / fill_s / 1e9 / HBM_PEAK_GBS) if lean_on and fill_s > 0 else None,
            "fill_time_note": ("lean stage: a chunk's forward launch overlaps the next chunk's reversed-graph launch in time; the fractions "
                               "are taken against min(sum of the launches' durations, the timed region) = %.2f ms per step" % (fill_s / args.steps * 1e3))
                              if lean_on and not lean_fused else None,
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
    if plain_stage:
        out["plain_stage"] = plain_stage
    if shortcut:
        out["exact_shortcut"] = shortcut
    if collective:
        out["dist"]["collective_ab"] = collective
    out["dist"]["per_rank"] = per_rank
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
    if args.sites_cpu_leg:
        sites_cpu_leg_main(args)
        return 0
    if args.config5_cpu_leg:
        config5_cpu_leg_main(args)
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["PG_BENCH_LAUNCHER"] = "bench.py --gpus %d (self-spawned torch.distributed.run)" % args.gpus
        return spawn_ranks(args)
    return main_rank(args)


if __name__ == "__main__":
    sys.exit(main())
