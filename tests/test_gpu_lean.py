"""The lean gssw stage (pg_ctx_set_lean, include/paragraph_amd.h): GraphAligner::alignRead(AF_ALL) (src/c++/lib/grm/GraphAligner.cpp:
308-404) from the reversed-graph fills of both strands, the forward-graph fill of the higher-scoring strand and -- only where the
record can depend on it -- that of the other strand.  Checked against the reference's own gssw.c on every field of the reference's
Read, and against the plain stage of this library on every field of pg_result (multi_mask: the bits of the fills that ran)."""
import random

import pytest

from tests import fuzzgen

pytestmark = pytest.mark.gpu

KEYS = ("graph_pos", "score", "mapq", "unique", "returned_reverse", "cigar", "clipped", "status")
REF_KEYS = ("graph_pos", "score", "mapq", "unique", "returned_reverse", "cigar")


def run(ctx, graphs, reads, gor, lean, flags=0xFFFFFFFF, before_align=None):
    from paragraph_amd import capi
    ctx.set_lean(2 if lean else 0)
    try:
        G = ctx.upload_graphs(graphs)
        b = ctx.new_batch()
        b.upload(G, reads, gor)
        if before_align:
            before_align(G, b)
        b.align(flags)
        res, ops = b.download()
        out = capi.results_to_dicts(res, ops)
        b.close()
        G.close()
    finally:
        ctx.set_lean(2)  # (the fixture's setting)
    return out


def compare(want, got, reads, what=""):
    """want: the plain stage; got: the lean one.  -> reads whose other strand's forward fill was skipped"""
    skipped = 0
    for i, (w, g) in enumerate(zip(want, got)):
        where = (what, i, reads[i], w, g)
        assert all(w[k] == g[k] for k in KEYS), where
        assert w["strand_score"] == g["strand_score"], where
        assert w["multi"][2:] == g["multi"][2:], where
        s = 1 if g["returned_reverse"] else 0
        if w["status"] == 0:
            assert w["multi"][s] == g["multi"][s], where
        if g["other_fwd_skipped"]:
            skipped += 1
            assert g["multi"][1 - s] == 0, where
        elif w["status"] == 0:
            assert w["multi"] == g["multi"], where
    return skipped


def compare_ref(want, got, reads, what=""):
    for i, (w, g) in enumerate(zip(want, got)):
        if w["score"] == 0:
            assert g["score"] == 0 and g["cigar"] == "" and g["graph_pos"] == 0 and g["status"] == 1 and g["mapq"] == w["mapq"], (what, i, reads[i], w, g)
        else:
            assert all(g[k] == w[k] for k in REF_KEYS) and g["status"] == 0, (what, i, reads[i], w, g)


def test_config2_reads(gpu_ctx, checker):
    from paragraph_amd import synth
    site, reads = synth.config2_reads(20000, read_len=150, seed=11)
    graphs = [(site.seqs, site.edges)]
    want = run(gpu_ctx, graphs, reads, None, False)
    got = run(gpu_ctx, graphs, reads, None, True)
    skipped = compare(want, got, reads, "config2")
    assert skipped > len(reads) // 2  # (most reads need three fills)
    ref = checker.align_batch(site.seqs, site.edges, reads[:3000], threads=8)
    compare_ref(ref, got[:3000], reads[:3000], "config2 vs reference")


def test_fuzz_many_graphs(gpu_ctx, checker):
    """the adversarial generator of tests/test_gpu_parity.py: near-identical branches, repeats, short nodes, N, indels -- many
    reads whose strands tie or are multi"""
    graphs, reads, gor, ref = [], [], [], []
    for gi, (seqs, edges, rs) in enumerate(fuzzgen.cases(777, 400, 10)):
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        ref.extend(checker.align_batch(seqs, edges, rs))
    want = run(gpu_ctx, graphs, reads, gor, False)
    got = run(gpu_ctx, graphs, reads, gor, True)
    compare(want, got, reads, "fuzz")
    compare_ref(ref, got, reads, "fuzz vs reference")
    assert sum(1 for g in got if not g["other_fwd_skipped"] and g["status"] == 0) > 50   # reads that took the fourth fill
    assert sum(1 for g in got if not g["unique"]) > 100


def test_fuzz_read_lengths(gpu_ctx, checker):
    """every byte variant (rows per lane 2 .. 16), and reads of 251+ bases in the same batch (their chunks run the plain stage)"""
    rng = random.Random(fuzzgen.salted(909))
    graphs, reads, gor, ref = [], [], [], []
    for gi in range(90):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=150, max_nodes=5)
        top = rng.choice([32, 64, 96, 128, 160, 192, 224, 250, 300])
        rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=max(1, top - 31), max_len=top)[:top] for _ in range(9)]
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        ref.extend(checker.align_batch(seqs, edges, rs, cigar_stride=2048))
    want = run(gpu_ctx, graphs, reads, gor, False)
    got = run(gpu_ctx, graphs, reads, gor, True)
    compare(want, got, reads, "lengths")
    compare_ref(ref, got, reads, "lengths vs reference")


def test_behind_the_exact_shortcut(gpu_ctx):
    """an active mask made on the device (pg_batch_retire_exact_matches) in front of the lean stage: the forced records stay, the
    other reads get the lean stage's"""
    from paragraph_amd import capi, synth
    site, reads = synth.config2_reads(6000, read_len=150, seed=3)
    graphs = [(site.seqs, site.edges)]
    want = run(gpu_ctx, graphs, reads, None, False)

    def shortcut(G, b):
        G.build_path_index(32)
        b.path_align()
        b.retire_exact_matches()

    got = run(gpu_ctx, graphs, reads, None, True, flags=capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS,
              before_align=shortcut)
    forced = 0
    for i, (w, g) in enumerate(zip(want, got)):
        assert all(w[k] == g[k] for k in KEYS), (i, reads[i], w, g)
        forced += -1 in g["strand_score"]
    assert forced > 600


def test_counts_after_lean(gpu_ctx):
    """the stages behind the gssw stage read its records: the count path's table after the lean stage = after the plain one"""
    import numpy as np
    from paragraph_amd import synth
    site, reads = synth.config2_reads(8000, read_len=150, seed=21)
    tabs = []
    for lean in (False, True):
        gpu_ctx.set_lean(2 if lean else 0)
        try:
            G = gpu_ctx.upload_graphs([(site.seqs, site.edges)])
            G.set_labels([site.labels])
            b = gpu_ctx.new_batch()
            b.upload(G, reads)
            b.set_fragments(np.arange(len(reads), dtype=np.uint32) // 2)
            b.align()
            b.count(remove_nonuniq=True, bad_align_frac=0.8)
            tabs.append(np.array(b.download_counts()[0], copy=True))
            b.close()
            G.close()
        finally:
            gpu_ctx.set_lean(2)
    assert np.array_equal(tabs[0], tabs[1]) and int(tabs[0].sum()) > 0


def test_many_chunks_two_batches_and_three_regions(checker):
    """a workspace of 96 MiB cuts 24 000 config-2 reads + 3 000 fuzz reads into many chunks; two batch objects aligned back to back meet
    on the regions (the lean stage's forward launch, traceback and second look of one chunk run beside the next chunk's reversed-graph
    fills); with pg_ctx_set_fill_streams(2): three regions, the reversed-graph launches alternating between two streams.  Records equal
    the plain stage's, and the reference's on a sample."""
    from paragraph_amd import capi, synth
    site, reads = synth.config2_reads(24000, read_len=150, seed=78)
    graphs, gor = [(site.seqs, site.edges)], [0] * len(reads)
    reads = list(reads)
    for gi, (seqs, edges, rs) in enumerate(fuzzgen.cases(910, 300, 10)):
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi + 1] * len(rs))
    out = {}
    for mode, streams in (("plain", 1), ("lean", 1), ("lean3", 2)):
        ctx = capi.Context(0, workspace_bytes=96 << 20, fill_streams=streams)
        ctx.set_lean(0 if mode == "plain" else 2)
        G = ctx.upload_graphs(graphs)
        batches = [ctx.new_batch(), ctx.new_batch()]
        for b in batches:
            b.upload(G, reads, gor)
        for _ in range(2):
            for b in batches:
                b.align(capi.AF_ALL)
        out[mode] = [capi.results_to_dicts(*b.download()) for b in batches]
        for b in batches:
            b.close()
        G.close()
        ctx.close()
    for mode in ("lean", "lean3"):
        for k in (0, 1):
            skipped = compare(out["plain"][k], out[mode][k], reads, mode)
            assert skipped > len(reads) // 2
    sample = list(range(0, 24000, 12))
    want = checker.align_batch(site.seqs, site.edges, [reads[i] for i in sample], threads=8)
    compare_ref(want, [out["lean3"][1][i] for i in sample], [reads[i] for i in sample], "three regions vs reference")


def test_tiny_runs_and_masks(gpu_ctx, checker):
    """runs of one to five reads per graph on repeat-rich graphs (the instance slots of a run with one or two pairs: the other strands'
    instances share the X strands' item), single-read batches, and a host-made active mask in front of the lean stage"""
    import numpy as np
    from paragraph_amd import capi
    rng = random.Random(fuzzgen.salted(4711))
    graphs, reads, gor, ref = [], [], [], []
    for gi in range(400):
        mode = rng.choice(["homo", "period", "two", None])
        n = rng.randint(2, 5)
        seqs = [fuzzgen.rand_seq(rng, rng.randint(3, 40), mode) for _ in range(n)]
        edges = [(i, j) for i in range(n) for j in range(i + 1, n) if j == i + 1 or rng.random() < 0.4]
        rs = []
        for _ in range(rng.choice([1, 1, 2, 3, 4, 5])):
            r = fuzzgen.rand_read(rng, seqs, edges, min_len=4, max_len=60)
            if rng.random() < 0.3:
                r = r[:len(r) // 2] + r[:len(r) // 2][::-1]  # (scores the same on both strands' like as not)
            rs.append(r or "A")
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        ref.extend(checker.align_batch(seqs, edges, rs))
    want = run(gpu_ctx, graphs, reads, gor, False)
    got = run(gpu_ctx, graphs, reads, gor, True)
    compare(want, got, reads, "tiny runs")
    compare_ref(ref, got, reads, "tiny runs vs reference")
    assert sum(1 for g in got if not g["other_fwd_skipped"] and g["status"] == 0) > 30
    # one read
    one = run(gpu_ctx, graphs[:1], reads[:1], gor[:1], True)
    compare_ref(ref[:1], one, reads[:1], "one read")
    # a host-made mask: every third read inactive (its record stays what the upload left), the others through the lean stage
    mask = np.array([0 if i % 3 == 0 else 1 for i in range(len(reads))], dtype=np.uint8)
    masked = run(gpu_ctx, graphs, reads, gor, True, flags=capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS,
                 before_align=lambda G, b: b.set_active(mask))
    act = [i for i in range(len(reads)) if mask[i]]
    compare([want[i] for i in act], [masked[i] for i in act], [reads[i] for i in act], "masked")
