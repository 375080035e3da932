"""CPU-side checks of what bench.py and the full-size GPU tests are built from: the site-set generator (per-site streams:
any subset equals the same sites of the full set, the forked generator equals the serial one), the partition the ranks use,
the CPU baseline leg (threads and processes write the same records into shared buffers; they equal the per-read oracle
calls) and the field-by-field comparison that produces bench.py's `verified` object."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_site_streams_subsets_and_forked_generator():
    from paragraph_amd import synth
    full = synth.mixed_sites(90, seed=4, site_streams=True)
    part = synth.mixed_sites(90, seed=4, site_streams=True, indices=[3, 41, 89])
    for got, i in zip(part, (3, 41, 89)):
        assert got.site.seqs == full[i].site.seqs and np.array_equal(got.reads, full[i].reads)
        assert np.array_equal(got.fragment, full[i].fragment) and np.array_equal(got.is_reverse, full[i].is_reverse)
    forked = synth.mixed_sites_parallel(90, seed=4, procs=3)
    assert len(forked) == 90
    assert all(a.site.seqs == b.site.seqs and np.array_equal(a.reads, b.reads) for a, b in zip(forked, full))
    assert {s.site.kind for s in full} == {"del", "longdel", "ins"}


def test_site_partition_is_balanced_and_complete():
    import bench
    from paragraph_amd import synth
    sites = synth.mixed_sites(400, seed=6, site_streams=True)
    for world in (1, 2, 8):
        sset = bench.SiteSet(sites, 150, world)
        seen = np.concatenate(sset.parts)
        assert sorted(seen.tolist()) == list(range(400))
        loads = [sset.weights[p].sum() for p in sset.parts]
        assert max(loads) <= 1.02 * sum(loads) / world
        arr, gor, frag, rev = sset.rank_arrays(world - 1)
        assert len(arr) == len(gor) == len(frag) == len(rev) == int(sset.n_reads_site[sset.parts[-1]].sum()) == sset.rank_reads(world - 1)
        assert set(np.unique(gor).tolist()) <= set(sset.parts[-1].tolist())


def test_hot_site_is_split_by_fragment_not_by_read():
    """A site at grmpy's read cap goes over all ranks by fragment id: every read lands on exactly one rank, mates on the
    same one, the other sites stay whole (ReadCounting.cpp:52-94, Fragment.cpp:141-181)."""
    import bench
    from paragraph_amd import synth
    sites = synth.mixed_sites(60, seed=6, site_streams=True) + synth.mixed_sites(1, seed=11, depth=1500.0, site_streams=True)
    hot = len(sites) - 1
    assert len(sites[hot].reads) >= 5000
    for world in (2, 8):
        sset = bench.SiteSet(sites, 150, world, split_reads=5000)
        assert sset.hot == [hot]
        assert sorted(np.concatenate(sset.parts).tolist()) == list(range(hot))
        owner_of_fragment = {}
        n_hot = 0
        total = 0
        for r in range(world):
            arr, gor, frag, rev = sset.rank_arrays(r)
            total += len(arr)
            sel = gor == hot
            n_hot += int(sel.sum())
            assert sel.sum() > 0
            for f in np.unique(frag[sel]).tolist():
                assert owner_of_fragment.setdefault(f, r) == r  # a fragment's reads never meet on two ranks
            whole = set(np.unique(gor[~sel]).tolist())
            assert whole <= set(sset.parts[r].tolist())
        assert n_hot == len(sites[hot].reads) and total == int(sset.n_reads_site.sum())
        assert len(owner_of_fragment) == len(np.unique(sites[hot].fragment))
        loads = [sset.rank_weight(r) for r in range(world)]
        assert max(loads) <= 1.05 * sum(loads) / world
    one = bench.SiteSet(sites, 150, 1)
    assert one.hot == [] and len(one.rank_arrays(0)[0]) == int(one.n_reads_site.sum()) == len(one.all_arrays()[0])


def test_cpu_leg_and_verification(tmp_path):
    """bench.py --cpu-leg on 3 000 config-2 reads: its records equal per-read oracle calls; verify_against_reference accepts a
    matching pg_result table built from them and counts an edited one."""
    import bench
    from oracle import oracle as orc
    from paragraph_amd import capi, synth
    site, arr = synth.config2_reads_packed(3000, read_len=150, seed=2)
    reads_file, out_file = tmp_path / "reads.npy", tmp_path / "cpu.npz"
    np.save(reads_file, arr)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-leg", "--cpu-reads-file", str(reads_file), "--cpu-out",
                        str(out_file), "--cpu-seconds", "0.5"], stdout=subprocess.PIPE, check=True)
    base = json.loads(p.stdout.decode().splitlines()[-1])
    assert base["unit"] == "reads/s" and base["value"] > 0 and base["cores"] >= 1 and base["kind"] in ("reference", "port")
    assert {"threads_one_process", "process_per_chunk", "probes", "sample"} <= set(base)
    z = np.load(out_file)
    res, cig = z["res"], z["cig"]
    assert len(res) == len(cig) >= 2000 and cig.shape[1] == bench.CIGAR_STRIDE
    from oracle import select
    chk = select.gssw()
    reads = [row.tobytes().decode() for row in arr[:40]]
    for i, w in enumerate(chk.align_batch(site.seqs, site.edges, reads)):
        assert w["cigar"].encode() == bytes(cig[i]).split(b"\0")[0] and w["graph_pos"] == res[i]["graph_pos"] and w["score"] == res[i]["score"]

    # a pg_result / pg_op table that says the same thing (host-only: pg_render_cigars needs no device)
    n = 300
    gres = np.zeros(n, dtype=capi.RESULT_DTYPE)
    ops = []
    codes = {c: k for k, c in enumerate(capi.OP_CHARS)}
    import re
    for i in range(n):
        gres[i]["graph_pos"], gres[i]["score"], gres[i]["mapq"] = res[i]["graph_pos"], res[i]["score"], res[i]["mapq"]
        gres[i]["is_unique"], gres[i]["returned_reverse"] = res[i]["unique"], res[i]["returned_reverse"]
        gres[i]["multi_mask"] = sum(int(res[i]["multi"][k]) << k for k in range(4))
        gres[i]["status"] = 1 if res[i]["score"] == 0 else 0
        gres[i]["ops_off"] = len(ops)
        text = bytes(cig[i]).split(b"\0")[0].decode()
        for node, body in re.findall(r"(\d+)\[([^\]]*)\]", text):
            for ln, op in re.findall(r"(\d+)([MXNIDS])", body):
                ops.append((int(node) << 16) | (codes[op] << 12) | int(ln))
        gres[i]["n_ops"] = len(ops) - gres[i]["ops_off"]
    ops = np.array(ops, dtype=np.uint32)
    v = bench.verify_against_reference(capi, gres, ops, res[:n], cig[:n])
    assert v == {"reads": n, "mismatches": 0, "fields": v["fields"], "forward_fills_of_the_other_strand_not_run": 0}
    # a record of the lean stage: the forward fill of the strand that was not returned did not run (bit 4), its bit reads 0
    k = next(i for i in range(n) if res[i]["score"] > 0)
    other = 0 if gres[k]["returned_reverse"] else 1
    keep = int(gres[k]["multi_mask"])
    gres[k]["multi_mask"] = (keep & ~(1 << other)) | 0x10
    v = bench.verify_against_reference(capi, gres, ops, res[:n], cig[:n])
    assert v["mismatches"] == 0 and v["forward_fills_of_the_other_strand_not_run"] == 1
    gres[k]["multi_mask"] = (keep ^ (1 << (1 - other))) | 0x10  # ... but the returned strand's own flag still counts
    assert bench.verify_against_reference(capi, gres, ops, res[:n], cig[:n])["mismatches"] == 1
    gres[k]["multi_mask"] = keep
    gres[7]["graph_pos"] += 1
    ops[int(gres[9]["ops_off"])] ^= 1  # one CIGAR element one base longer / shorter
    v = bench.verify_against_reference(capi, gres, ops, res[:n], cig[:n])
    assert v["mismatches"] == 2 and v["first_mismatch"]["read"] == 7


def test_config5_cpu_leg(tmp_path):
    """bench.py --config5-cpu-leg (the checker side of the configs[4] leg) on two small long-node graphs: its records equal
    per-read checker calls on the same reads."""
    import pickle
    import bench
    from oracle import select
    cases = bench.config5_cases(2, 3)
    assert all(a.shape == (3, bench.CONFIG5_READ_LEN) and 2000 <= len(s.seqs[2]) <= 8000 for s, a in cases)
    in_file, out_file = tmp_path / "c5.pkl", tmp_path / "ref.pkl"
    with open(in_file, "wb") as f:
        pickle.dump([(s, a[:2]) for s, a in cases], f, protocol=4)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config5-cpu-leg", "--cpu-reads-file", str(in_file), "--cpu-out",
                        str(out_file)], stdout=subprocess.PIPE, check=True)
    info = json.loads(p.stdout.decode().splitlines()[-1])
    assert info["graphs"] == 2 and info["reads"] == 4
    with open(out_file, "rb") as f:
        want = pickle.load(f)
    chk = select.gssw()
    for (s, a), (res, cig) in zip(cases, want):
        assert cig.shape == (2, bench.CONFIG5_CIGAR_STRIDE)
        for i, w in enumerate(chk.align_batch(s.seqs, s.edges, [row.tobytes().decode() for row in a[:2]])):
            assert w["cigar"].encode() == bytes(cig[i]).split(b"\0")[0] and w["score"] == res[i]["score"] > 100


def test_bench_spawn_command(monkeypatch):
    """--gpus N without WORLD_SIZE: bench.py re-executes itself under torch.distributed.run on 127.0.0.1 with N ranks."""
    import bench
    seen = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--workload", "config3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main() == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--workload", "config3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "PG_BENCH_LAUNCHER" in os.environ


def test_e2e_data_set_two_parts_and_reference_view(tmp_path, monkeypatch):
    """bench.py's e2e leg, the CPU side: two 'ranks' each make the pieces of their half of the sites, rank 0 joins them into the
    same data set one process makes alone; the sampled sites come out as the reference's workflow would see them (GraphInput's
    one-base source / sink nodes, the reads ReadExtraction keeps, BAM strand flags) and the reference's counting finds the
    simulated alleles on them."""
    import argparse
    import filecmp
    import bench
    from paragraph_amd import synth_e2e
    one = synth_e2e.make_dataset(str(tmp_path / "one"), n_sites=24, seed=1, procs=2, keep_sites=[0, 8, 16])
    args = argparse.Namespace(e2e_sites=24, e2e_verify=3, no_cpu_baseline=False, read_len=150)
    monkeypatch.setenv("MASTER_PORT", "t%d" % os.getpid())
    monkeypatch.setattr(bench.tempfile, "gettempdir", lambda: str(tmp_path))
    monkeypatch.setattr(bench.os.path, "isdir", lambda p: False if p == "/dev/shm" else os.path.exists(p) and not os.path.isfile(p))
    import threading
    got = {}
    t = threading.Thread(target=lambda: got.setdefault(1, bench.prepare_e2e(args, 1, 2, 2)))
    t.start()
    got[0] = bench.prepare_e2e(args, 0, 2, 2)
    t.join()
    for name in ("reads.bam", "reads.bam.bai", "ref.fa", "truth.json"):
        assert filecmp.cmp(os.path.join(got[0]["dir"], name), os.path.join(str(tmp_path / "one"), name), shallow=False), name
    assert got[0]["reads"] == got[1]["reads"] == one["reads"] and got[0]["sample_idx"] == [0, 8, 16]
    assert got[1]["truth"] == got[0]["truth"] and got[1]["graphs"] == got[0]["graphs"]
    sample = bench.e2e_reference_sites(got[0], 150)
    from oracle import select
    chk = select.gssw()
    for i, s in zip(got[0]["sample_idx"], sample):
        assert s.site.seqs[0] == s.site.seqs[-1] == "X" and 150 < len(s.reads) < len(got[0]["kept"][i]["pos"])
        w = bench.reference_site_outcome(chk, s, bench.SITES_CIGAR_STRIDE)
        counts = {(s.site.names[a], s.site.names[b]): int(w["edge_counts"][k][0]) for k, (a, b) in enumerate(s.site.edges)}
        alt_edges = [e for e, labs in ((e, s.site.labels.get((s.site.names.index(e[0]), s.site.names.index(e[1])), [])) for e in counts) if labs == ["ALT"]]
        ref_edges = [e for e, labs in ((e, s.site.labels.get((s.site.names.index(e[0]), s.site.names.index(e[1])), [])) for e in counts) if labs == ["REF"]]
        gt = got[0]["truth"][i]["gt"]
        assert (sum(counts[e] for e in alt_edges) > 0) == ("ALT" in gt) and (sum(counts[e] for e in ref_edges) > 0) == ("REF" in gt)


def test_roofline_head_names_the_bound_that_binds():
    import bench
    with_counters = bench.roofline_head({"traffic": 5.0e10, "valu": {"issue_frac_with_measured_pairing": 0.77, "issue_frac": 0.74, "clock_ghz": 2.4}}, 9900.0)
    assert with_counters["bound"] == "valu" and with_counters["frac"] == 0.77 and with_counters["traffic"] == 5.0e10
    assert abs(with_counters["achieved"] / with_counters["peak"] - 0.77) < 1e-12 and with_counters["hbm_formula_frac"] > 1.0
    without = bench.roofline_head({"traffic": None, "valu": {"why": "no counters"}}, 9900.0)
    assert without["bound"] == "hbm" and without["unit"] == "GB/s" and without["frac"] == without["hbm_formula_frac"]
