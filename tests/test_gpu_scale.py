"""Full-size parity on the GPU: the BASELINE.json configurations at sizes the driver's bench runs, against the reference's
own gssw.c / graph-tools code (oracle/_ref; the plain-C port where it is absent).  The reference side is computed by
tests/scale_oracle.py in a process of its own, fanned out over the host's cores; comparisons are vectorised.

  configs[1]  262 144 config-2 reads                      (every pg_result field + CIGAR string)
  configs[2]  2 000 mixed DEL / long-DEL / INS sites, 30x  (alignments, per-read supports, per-site count tables)
  configs[4]  2 400 reads of 250 bp on 2-8 kb ALT nodes
  stress      tests/stress_parity.py, two salts            (adversarial graphs, word mode, far predecessors)
  N > 1       bench.py --gpus 2 --workload config3 on this box (ranks share the GPU, gloo): reduced table == 1-rank table"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _upload_and_align(ctx, graphs, arr, gor=None):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    n, L = arr.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
    b = ctx.new_batch()
    b.upload(G, (off, np.ascontiguousarray(arr).reshape(-1)), gor)
    b.align(capi.AF_ALL)
    res, ops = b.download()
    return G, b, res, ops


def test_config2_262144_reads(gpu_ctx):
    import bench
    from paragraph_amd import capi, synth
    from tests import scale_oracle
    n = 262144
    site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
    ref_res, ref_cig = scale_oracle.run("config2", n, 2)
    G, b, res, ops = _upload_and_align(gpu_ctx, [(site.seqs, site.edges)], arr)
    v = bench.verify_against_reference(capi, res, ops, ref_res, ref_cig)
    b.close()
    G.close()
    assert v["reads"] == n and v["mismatches"] == 0, v


def test_config3_2000_sites(gpu_ctx):
    import bench
    from paragraph_amd import capi, synth
    from tests import scale_oracle
    n_sites = 2000
    sites = synth.mixed_sites(n_sites, seed=5, site_streams=True)
    want = scale_oracle.run("config3", n_sites, 5)
    arr = np.concatenate([s.reads for s in sites])
    gor = np.concatenate([np.full(len(s.reads), i, dtype=np.uint32) for i, s in enumerate(sites)])
    G, b, res, ops = _upload_and_align(gpu_ctx, [(s.site.seqs, s.site.edges) for s in sites], arr, gor)
    G.set_labels([s.site.labels for s in sites])
    b.set_fragments(np.concatenate([s.fragment for s in sites]), np.concatenate([s.is_reverse for s in sites]))
    b.count(remove_nonuniq=True)
    table, sup, path = b.download_counts()
    # ---- alignments, all reads at once
    ref_res = np.concatenate([w["res"] for w in want])
    ref_cig = np.concatenate([w["cig"] for w in want])
    v = bench.verify_against_reference(capi, res, ops, ref_res, ref_cig)
    assert v["reads"] == len(arr) > 400000 and v["mismatches"] == 0, v
    # ---- per-read outcome of the count path, all reads at once
    ref_status = np.concatenate([w["status"] for w in want])
    ref_labels = np.concatenate([w["label_mask"] for w in want])
    assert np.array_equal(sup["status"], ref_status), np.nonzero(sup["status"] != ref_status)[0][:5]
    mapped = ref_status == 1
    assert np.array_equal(sup["label_mask"][mapped], ref_labels[mapped])
    # ---- node / edge supports read by read on every 16th site, count tables on every site
    cnt = capi.decode_counts(G, table)
    k = 0
    kinds = set()
    for si, (s, w) in enumerate(zip(sites, want)):
        kinds.add(s.site.kind)
        n = len(s.reads)
        if si % 16 == 0:
            ds = capi.decode_supports(G, gor[k:k + n], sup[k:k + n], path)
            for i in range(n):
                if w["status"][i] == 1:
                    assert ds[i]["nodes"] == w["nodes"][i] and ds[i]["edges"] == w["edges"][i], (si, i)
        c = cnt[si]
        assert np.array_equal(c["node_counts"], w["node_counts"]), (si, s.site.kind)
        for ei, e in enumerate(s.site.edges):
            assert c["edge_counts"][tuple(e)] == [int(x) for x in w["edge_counts"][ei]], (si, e)
        assert c["seq_counts"] == w["seq_counts"], (si, c["seq_counts"], w["seq_counts"])
        k += n
    b.close()
    G.close()
    assert kinds == {"del", "longdel", "ins"}


def test_config5_2400_long_node_reads(gpu_ctx):
    import bench
    from paragraph_amd import capi
    from tests import scale_oracle
    cases = scale_oracle.config5_cases(50, 48, 9)
    want = scale_oracle.run("config5", 50, 48, 9)
    arr = np.concatenate([a for _, a in cases])
    gor = np.concatenate([np.full(len(a), i, dtype=np.uint32) for i, (_, a) in enumerate(cases)])
    G, b, res, ops = _upload_and_align(gpu_ctx, [(s.seqs, s.edges) for s, _ in cases], arr, gor)
    v = bench.verify_against_reference(capi, res, ops, np.concatenate([w["res"] for w in want]), np.concatenate([w["cig"] for w in want]))
    b.close()
    G.close()
    assert v["reads"] == 2400 and v["mismatches"] == 0, v


@pytest.mark.parametrize("seed", [1, 2])
def test_stress_parity(gpu_ctx, checker, seed):
    from tests import fuzzgen, stress_parity
    total = stress_parity.run(1500, fuzzgen.salted(seed), gpu_ctx, checker, verbose=False)
    assert total > 9000


def test_bench_two_ranks_shard_one_site_set():
    """configs[3] on this box: bench.py spawns 2 ranks itself; with one GPU they share it and reduce over gloo.  The reduced
    counter table must equal the table of a 1-rank pass over all sites."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "config3", "--sites", "400",
                        "--steps", "2", "--warmup", "1", "--workspace-gib", "16", "--e2e-sites", "300", "--e2e-steps", "1", "--e2e-verify", "40"],
                       stdout=subprocess.PIPE, env=env, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["dist"]["world"] == 2
    assert out["sites"]["reduce_equals_single"] is True, out["sites"]
    assert out["sites"]["sites"] == 400 and len(out["sites"]["shard_reads"]) == 2 and min(out["sites"]["shard_reads"]) > 0
    assert out["sites"]["tallies"]["aligned"] == out["sites"]["reads"]
    # BAM -> genotypes, two ranks: each genotypes its half of the sites from the one BAM, the edge-count table is summed over gloo
    e = out["e2e"]
    assert e["sites"] == 300 and e["mismatches"] == 0 and e["genotypes_equal_truth"] >= 298 and e["documents_with_error"] == 0, e
    assert len(e["per_rank"]) == 2 and sorted(r["sites"] for r in e["per_rank"]) == [150, 150]
    assert e["edge_table"]["reduced_equals_own_on_own_sites"] is True and e["edge_table"]["sum"] > 300 * 20
    assert e["verified"]["sites"] == 40 and e["verified"]["site_mismatches"] == 0 and e["verified"]["reads"] > 40 * 150, e["verified"]
    assert e["sites_genotyped_per_s"] > 0 and e["cpu_us_per_site_sample"] > 0


def _run_bench(argv, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, stdout=subprocess.PIPE, env=env, timeout=timeout)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    return json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])


def test_bench_two_ranks_split_a_hot_site_by_fragment():
    """a hot site (here ~10 000 reads, grmpy's cap) is split over the ranks by FRAGMENT id -- mates stay
    together -- and the all-reduce then really sums that site's counters: reduced table == 1-rank table."""
    out = _run_bench(["--gpus", "2", "--workload", "config3", "--sites", "200", "--hot-site-depth", "1500", "--steps", "2",
                      "--warmup", "1", "--workspace-gib", "16", "--e2e-steps", "0"])
    hot = out["sites"]["hot_sites_split_by_fragment"]
    assert len(hot) == 1 and hot[0]["reads"] >= 5000 and min(hot[0]["reads_per_rank"]) > 1000, hot
    assert out["sites"]["reduce_equals_single"] is True, out["sites"]
    assert out["sites"]["tallies"]["aligned"] == out["sites"]["reads"]


def test_bench_eight_ranks_share_the_gpu():
    """The launch the driver's 8-GPU box will see, on the one GPU: eight ranks (sharing the device, so the reduce hops through
    the host under gloo -- with eight devices the same code takes the nccl branch and fails unless every rank has its own).
    configs[3] with a hot site: 8 non-empty balanced shards, the hot site split 8 ways by fragment, reduced table == 1-rank
    table, a sample of rank 0's sites equal to the reference's code; then the default weak leg (config 2 per rank + the sites
    leg).  Bounded wall clock on the box's CPU quota (the generators are capped by the cgroup quota, not by the CPUs visible)."""
    import time
    t0 = time.perf_counter()
    out = _run_bench(["--gpus", "8", "--workload", "config3", "--sites", "800", "--hot-site-depth", "1500", "--steps", "2",
                      "--warmup", "1", "--workspace-gib", "8", "--sites-verify", "100", "--e2e-sites", "800", "--e2e-steps", "1",
                      "--e2e-verify", "50"], timeout=1500)
    d, s = out["dist"], out["sites"]
    assert out["n_gpus"] == 8 and d["world"] == 8 and d["shared_device"] is True and d["backend"] == "gloo", d
    assert d["placement_ok"] is True and len(d["ranks"]) == 8 and sorted(r["rank"] for r in d["ranks"]) == list(range(8))
    assert s["reduce_equals_single"] is True, s
    assert len(s["shard_reads"]) == 8 and min(s["shard_reads"]) > 0 and s["shard_imbalance"] < 1.05, s
    hot = s["hot_sites_split_by_fragment"]
    assert len(hot) == 1 and len(hot[0]["reads_per_rank"]) == 8 and min(hot[0]["reads_per_rank"]) > 500, hot
    assert s["verified"]["sites"] == 100 and s["verified"]["mismatches"] == 0 and s["verified"]["reads"] > 10000, s["verified"]
    assert len(s["per_rank"]) == 8 and all(r["fill_launches"] > 0 for r in s["per_rank"]), s["per_rank"]
    assert s["collective_ab"]["with_vs_without"] > 0
    assert s["tallies"]["aligned"] == s["reads"]
    # the BAM -> genotypes leg with eight ranks: every rank its eighth of the sites from the one BAM the eight made together
    e = out["e2e"]
    assert e["sites"] == 800 and e["mismatches"] == 0 and e["genotypes_equal_truth"] >= 796, e
    assert len(e["per_rank"]) == 8 and all(r["sites"] == 100 for r in e["per_rank"]) and e["verified"]["site_mismatches"] == 0
    # the default workload at N = 8: weak scaling of config 2 + the sites leg, per-rank fill times in the line
    out = _run_bench(["--gpus", "8", "--reads", "100000", "--steps", "2", "--warmup", "1", "--sites", "400", "--sites-steps", "1",
                      "--sites-verify", "50", "--workspace-gib", "8", "--e2e-steps", "0"], timeout=1500)
    d = out["dist"]
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and d["world"] == 8 and len(d["per_rank"]) == 8, d
    assert all(r["fill_launches"] > 0 and r["fill_ms_per_launch"] > 0 for r in d["per_rank"])
    assert out["counts"]["tallies"]["aligned"] == 8 * 100000, out["counts"]
    assert out["sites"]["reduce_equals_single"] is True and out["sites"]["verified"]["mismatches"] == 0, out["sites"]
    assert d["collective_ab"]["with_vs_without"] > 0
    assert time.perf_counter() - t0 < 900


def test_bench_one_rank_runs_the_rccl_reduce_stream_ordered():
    """N = 1 with a world-size-1 RCCL communicator: the all-reduce of the counter table is inside every timed step, ordered
    by events (no host synchronisation), and costs (next to) nothing; the counts are those of the plain path."""
    out = _run_bench(["--reads", "400000", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--stream-batches", "0",
                      "--sites", "300", "--sites-steps", "2", "--e2e-sites", "600", "--e2e-steps", "2"])
    d = out["dist"]
    assert d["backend"] == "nccl" and d["world"] == 1 and d["collective_in_step"] is True, d
    assert d["ranks"][0]["device"] == 0
    ab = d["collective_ab"]
    assert ab["with_vs_without"] < 1.05, ab
    t = out["counts"]["tallies"]
    assert t["aligned"] == 400000, t
    assert out["sites"]["tallies"]["aligned"] == out["sites"]["reads"]
    e = out["e2e"]  # one rank, no CPU checker leg (--no-cpu-baseline): genotypes against the truth, the table through RCCL
    assert e["sites"] == 600 and e["mismatches"] == 0 and e["genotypes_equal_truth"] >= 597 and "verified" not in e, e
    assert e["edge_table"]["reduce_ms"] is not None and e["host_threads"] >= 1
    plain = _run_bench(["--reads", "400000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--stream-batches", "0",
                        "--sites-steps", "0", "--collective", "off", "--e2e-steps", "0"])
    assert plain["dist"]["collective_in_step"] is False
    assert plain["counts"] == out["counts"]


def test_count_stream_events_order_a_foreign_stream():
    """pg_ctx_count_record / pg_ctx_count_wait with the runtime's own event handles (here torch's): a foreign stream reads
    the counter table behind the count kernels, and the next zeroing waits for that reader -- no host synchronisation.  In a
    process of its own: torch brings its own HIP runtime, which has to be loaded before the library's (as in bench.py)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "count_events_check.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"count events ok" in p.stdout, p.stdout.decode()[-3000:]
