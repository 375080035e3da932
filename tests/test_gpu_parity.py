"""GPU parity: the HIP path (through the C ABI) must be bit-exact with the oracle on graph_pos, CIGAR,
score, MAPQ, uniqueness, strand and the four multi flags."""
import pytest

from tests import fuzzgen

pytestmark = pytest.mark.gpu

KEYS = ("graph_pos", "score", "mapq", "unique", "returned_reverse", "multi", "cigar")

ALIGNS_GRAPH = (["AAAAAAAAAAA", "TTTTTTTT", "GGGGGGGG", "AAAAAAAAAAA"], [(0, 1), (0, 2), (0, 3), (1, 3), (2, 3)])
ALIGNS_READS = ["AAAAAAAATTTTCTTTAAAAAAAA", "TTTTTTAAAGAAAATTTTTTT", "AAAAAGCGGGGGGAAAAAA", "AAAAGCGGGGGGAAAAAA",
                "TTTTTTCCCCCCGCTTTTT", "AAAAAAAAAAAAAAAAAAA"]
# src/c++/test/test_paragraph_parts.cpp:111-143 (graphPos, graphCigar, score, mapq, reverse strand)
ALIGNS_EXPECTED = [(3, "0[8M]1[4M1X3M]3[8M]", 19, 60, False), (4, "0[7M]1[4M1X3M]3[6M]", 16, 60, True),
                   (6, "0[5M]2[1M1X6M]3[6M]", 14, 60, False), (7, "0[4M]2[1M1X6M]3[6M]", 13, 60, False),
                   (6, "0[5M]2[1M1X6M]3[6M]", 14, 60, True), (0, "0[11M]3[8M]", 19, 60, False)]


def gpu_align(ctx, graphs, reads, graph_of_read=None, flags=0xFFFFFFFF):
    from paragraph_amd import capi
    G = ctx.upload_graphs(graphs)
    b = ctx.new_batch()
    b.upload(G, reads, graph_of_read)
    b.align(flags)
    res, ops = b.download()
    out = capi.results_to_dicts(res, ops)
    b.close()
    G.close()
    return out


def compare(got, want, reads, what=""):
    bad = []
    for i, (g, w) in enumerate(zip(got, want)):
        if w["score"] == 0:
            # degenerate all-zero fill: empty CIGAR at position 0; flagged with status 1
            ok = g["score"] == 0 and g["cigar"] == "" and g["graph_pos"] == 0 and g["status"] == 1 \
                and fuzzgen.multi_equal(g, w) and g["mapq"] == w["mapq"]
        else:
            ok = all(g[k] == w[k] for k in KEYS if k != "multi") and fuzzgen.multi_equal(g, w) and g["status"] == 0
        if not ok:
            bad.append((i, reads[i], g, w))
    assert not bad, "%s: %d/%d mismatches, first: %r" % (what, len(bad), len(reads), bad[:2])


def test_reference_unit_vectors(gpu_ctx):
    got = gpu_align(gpu_ctx, [ALIGNS_GRAPH], ALIGNS_READS)
    for g, (pos, cigar, score, mapq, rev) in zip(got, ALIGNS_EXPECTED):
        assert (g["graph_pos"], g["cigar"], g["score"], g["mapq"], g["returned_reverse"]) == (pos, cigar, score, mapq, rev)


def test_fuzz_many_graphs(gpu_ctx, checker):
    graphs, reads, gor = [], [], []
    want = []
    for gi, (seqs, edges, rs) in enumerate(fuzzgen.cases(2024, 300, 10)):
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(checker.align_batch(seqs, edges, rs))
    got = gpu_align(gpu_ctx, graphs, reads, gor)
    compare(got, want, reads, "fuzz")


def test_fuzz_long_reads(gpu_ctx, checker):
    import random
    rng = random.Random(fuzzgen.salted(5))
    graphs, reads, gor, want = [], [], [], []
    for gi in range(60):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=150, max_nodes=5)
        rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=100, max_len=250)[:250] for _ in range(8)]
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(checker.align_batch(seqs, edges, rs))
    got = gpu_align(gpu_ctx, graphs, reads, gor)
    compare(got, want, reads, "fuzz-long")


def test_fuzz_word_mode_reads(gpu_ctx, checker):
    """251..512 bp reads: the wide kernel variants (two bytes of H per cell) = gssw's 16-bit word mode, incl. the
    reference's byte-pointer scan of word matrices in alignsEndAtMultNodes (scores 251..255 / >= 256)."""
    import random
    rng = random.Random(fuzzgen.salted(512))
    graphs, reads, gor, want = [], [], [], []
    for gi in range(120):
        if gi % 3 == 0:
            seqs, edges = fuzzgen.rand_graph(rng, max_len=260, max_nodes=5)
            rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=240, max_len=512)[:512] for _ in range(8)]
        else:
            seqs, edges, rs = fuzzgen.long_read_case(rng, 8)
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(checker.align_batch(seqs, edges, rs, cigar_stride=2048))
    got = gpu_align(gpu_ctx, graphs, reads, gor)
    compare(got, want, reads, "fuzz-word")
    assert sum(1 for w in want if w["score"] >= 251) > 300
    assert sum(1 for w in want if 251 <= max(w["scores"]) <= 255) > 15
    assert sum(1 for w in want if max(w["scores"]) >= 251 and any(w["multi"])) > 0
    assert max(len(r) for r in reads) > 480


def test_config2_sample(gpu_ctx, checker):
    from paragraph_amd import synth
    site, reads = synth.config2_reads(4096, read_len=150, seed=2)
    want = checker.align_batch(site.seqs, site.edges, reads, threads=8)
    got = gpu_align(gpu_ctx, [(site.seqs, site.edges)], reads)
    compare(got, want, reads, "config2")


def test_flags(gpu_ctx, checker):
    from paragraph_amd import capi
    seqs, edges = ALIGNS_GRAPH
    for flags in (capi.AF_CIGAR, capi.AF_CIGAR | capi.AF_BOTH_STRANDS, capi.AF_CIGAR | capi.AF_REVERSE_GRAPH):
        want = checker.align_batch(seqs, edges, ALIGNS_READS, flags=flags)
        got = gpu_align(gpu_ctx, [ALIGNS_GRAPH], ALIGNS_READS, flags=flags)
        compare(got, want, ALIGNS_READS, "flags=%d" % flags)


def test_empty_and_ragged(gpu_ctx, checker):
    seqs, edges = ALIGNS_GRAPH
    reads = ["", "A", "ACGT" * 60 + "AC", "", "TTTTTTTT"]
    got = gpu_align(gpu_ctx, [ALIGNS_GRAPH], reads)
    assert got[0]["status"] == 1 and got[3]["status"] == 1  # skipped like Align.cpp:74-77
    idx = [1, 2, 4]
    want = checker.align_batch(seqs, edges, [reads[i] for i in idx])
    compare([got[i] for i in idx], want, [reads[i] for i in idx], "ragged")


def test_length_boundaries(gpu_ctx, checker):
    """Every variant boundary: rows-per-lane steps (multiples of 32 / 64), byte vs wide (250 / 251), packed vs general path
    (512 / 513), and the general path's own limit (16 000 bases)."""
    import random
    rng = random.Random(fuzzgen.salted(1234))
    seqs, edges, _ = fuzzgen.long_read_case(rng, 1)
    path = seqs[0] + seqs[1] + seqs[-1]
    lens = [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 159, 160, 161, 191, 192, 193, 223, 224,
            225, 249, 250, 251, 252, 255, 256, 257, 319, 320, 321, 383, 384, 385, 447, 448, 449, 479, 480, 481, 511, 512, 513, 514, 640, 1025]
    reads = []
    for L in lens:
        st = rng.randrange(max(1, len(path) - L))
        r = (path[st:st + L] + fuzzgen.rand_seq(rng, L))[:L]
        reads.append(fuzzgen.mutate(rng, r, sub=0.01, indel=0.0)[:L] or "A")
        reads.append(r)
    want = checker.align_batch(seqs, edges, reads, cigar_stride=4096)
    got = gpu_align(gpu_ctx, [(seqs, edges)], reads)
    compare(got, want, reads, "length-boundaries")
    from paragraph_amd import capi
    with pytest.raises(Exception):
        gpu_align(gpu_ctx, [(seqs, edges)], ["A" * 16001])


def test_golden_fixtures(gpu_ctx):
    """Committed vectors generated from the reference's gssw.c (tests/golden/make_golden.py)."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.abspath(__file__))
    n = 0
    for path in sorted(glob.glob(os.path.join(root, "golden", "*.json"))):
        with open(path) as f:
            fx = json.load(f)
        graphs, reads, gor, want = [], [], [], []
        for gi, s in enumerate(fx["sets"]):
            graphs.append((s["nodes"], [tuple(e) for e in s["edges"]]))
            reads.extend(s["reads"])
            gor.extend([gi] * len(s["reads"]))
            want.extend(s["expected"])
        got = gpu_align(gpu_ctx, graphs, reads, gor)
        compare(got, want, reads, os.path.basename(path))
        for g, w in zip(got, want):
            assert g["strand_score"][0] == w["scores"][0] and g["strand_score"][1] == w["scores"][1]
        n += len(reads)
    assert n > 900


def test_host_cpp_mirror():
    """The reference-shaped C++ host classes (grm::alignReads, GraphAligner, CompositeAligner, SiteBatcher) on
    the reference's own unit-test fixtures: tests/host_cpp/test_host.cpp."""
    import os
    import subprocess
    from paragraph_amd import build
    exe = build.HOST_TEST
    if not os.path.exists(exe):
        build.build_host()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


def test_many_tiny_nodes(gpu_ctx, checker):
    """Graphs with hundreds of 1-4 bp nodes: single-column nodes (FIRST and LAST on the same column), long
    predecessor lists, seeds on every column, node-key table far larger than the profile."""
    import random
    rng = random.Random(fuzzgen.salted(4321))
    graphs, reads, gor, want = [], [], [], []
    for gi in range(6):
        n = rng.choice([60, 150, 300])
        seqs = [fuzzgen.rand_seq(rng, rng.randint(1, 4), "rand") for _ in range(n)]
        edges = set()
        for t in range(1, n):
            for f in rng.sample(range(max(0, t - 6), t), min(t, rng.randint(1, 3))):
                edges.add((f, t))
        edges = sorted(edges)
        rs = [fuzzgen.rand_read(rng, seqs, edges, min_len=20, max_len=140) for _ in range(12)]
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi] * len(rs))
        want.extend(checker.align_batch(seqs, edges, rs))
    got = gpu_align(gpu_ctx, graphs, reads, gor)
    compare(got, want, reads, "tiny-nodes")


def test_documented_limits_fail_loudly(gpu_ctx):
    """Every limit of the envelope (include/paragraph_amd.h) answers with PG_ERR_UNSUPPORTED -- never with a wrong result:
    65 536 nodes, 257 labels, 127 klib paths, a 16 001-base read.  (Round 3: more than 4 095 nodes, a direction longer than 65 519
    columns and reads of 513..16 000 bases are no longer limits -- they take the general path, tests/test_gpu_general.py.)"""
    from paragraph_amd import capi
    chain = (["A"] * 65536, [(i, i + 1) for i in range(65535)])
    with pytest.raises(capi.PgError) as e:
        gpu_ctx.upload_graphs([chain])
    assert e.value.status == 4 and "65535" in str(e.value)
    gpu_ctx.upload_graphs([(["ACGT"] * 4096, [(i, i + 1) for i in range(4095)])]).close()  # general path
    gpu_ctx.upload_graphs([(["A" * 40000, "C" * 25600], [(0, 1)])]).close()  # general path
    G = gpu_ctx.upload_graphs([ALIGNS_GRAPH])
    names = ["L%03d" % i for i in range(257)]
    with pytest.raises(capi.PgError) as e:
        G.set_labels([{(0, 1): names[:1]}], [names])  # 257 labels declared
    assert e.value.status == 4
    G.set_labels([{(0, 1): names[:256]}], [names[:256]])  # 256 labels are in (four words: tests/test_gpu_counts.py)
    G.set_labels([{(0, 1): names[:64]}], [names[:64]])
    with pytest.raises(capi.PgError) as e:
        G.build_klib_index([[[0, 1, 3]] * 127])
    assert e.value.status == 4
    G.build_klib_index([[[0, 1, 3]] * 126])
    b = gpu_ctx.new_batch()
    with pytest.raises(capi.PgError) as e:
        b.upload(G, ["A" * 16001])
    assert e.value.status == 4
    b.upload(G, ["A" * 512, "A" * 513])
    b.align()
    res, _ = b.download()
    assert res[0]["status"] & 0xFF in (0, 1)
    b.close()
    G.close()


@pytest.mark.parametrize("env", [{"PG_TRACE_BLOCKS": "0"}, {"PG_TRACE_BLOCKS": "7"}, {"PG_WIDE16": "1"}, {"PG_LEAN": "0"},
                                 {"PG_LEAN": "0", "PG_TRACE_BLOCKS": "0"}, {"PG_LEAN_FUSED": "0"}, {"PG_LEAN_FUSED": "0", "PG_LEAN_ONE_STREAM": "1"}])
def test_launch_settings_do_not_change_results(env):
    """PG_LEAN=0: the plain gssw stage (four fills per read, all four multi flags) instead of the lean one this file's contexts run
    by default; PG_LEAN_FUSED=0: the lean stage as three launches per chunk (reversed-graph fills, pick, forward-graph fills of instance items) instead of
    the one fused launch; with PG_LEAN_ONE_STREAM its forward launch on the fill stream.
    The traceback walks its work-item pairs in a grid-stride loop of a bounded number of wavefronts (PG_TRACE_BLOCKS; 0 = one
    wavefront per pair), and PG_WIDE16 selects the 16-lane kernels for reads of 251-512 bases: the settings are read once per
    process, so each one gets a process of its own running the read-length, word-mode and fuzz tests of this file."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-p", "no:cacheprovider", "-k",
                        "fuzz_many_graphs or fuzz_word_mode_reads or length_boundaries or many_tiny_nodes"],
                       cwd=root, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_two_fill_streams_three_regions(checker):
    """pg_ctx_set_fill_streams(2): three workspace regions, consecutive chunks' fills on two streams (the next chunk's wavefronts
    take the slots a draining launch leaves).  A workspace of 96 MiB cuts 24 000 config-2 reads + 3 000 reads of mixed fuzz graphs
    into many chunks; two batch objects are aligned back to back so that chunks of different batches meet on the regions.
    Records equal those of the one-stream context field for field, and the reference's on a sample."""
    import numpy as np
    from paragraph_amd import capi, synth
    site, reads = synth.config2_reads(24000, read_len=150, seed=77)
    graphs, gor = [(site.seqs, site.edges)], [0] * len(reads)
    reads = list(reads)
    for gi, (seqs, edges, rs) in enumerate(fuzzgen.cases(909, 300, 10)):
        graphs.append((seqs, edges))
        reads.extend(rs)
        gor.extend([gi + 1] * len(rs))
    out = {}
    for streams in (1, 2):
        ctx = capi.Context(0, workspace_bytes=96 << 20, fill_streams=streams)
        G = ctx.upload_graphs(graphs)
        batches = [ctx.new_batch(), ctx.new_batch()]
        for b in batches:
            b.upload(G, reads, gor)
        for _ in range(2):
            for b in batches:
                b.align(capi.AF_ALL)
        got = [b.download() for b in batches]
        out[streams] = [capi.results_to_dicts(r, o) for r, o in got]
        for b in batches:
            b.close()
        G.close()
        ctx.close()
    for k in (0, 1):
        for a, b in zip(out[1][k], out[2][k]):
            assert all(a[key] == b[key] for key in KEYS) and a["status"] == b["status"]
    sample = list(range(0, 24000, 12)) + list(range(24000, len(reads)))
    want = checker.align_batch(site.seqs, site.edges, [reads[i] for i in sample if i < 24000])
    compare([out[2][1][i] for i in sample if i < 24000], want, [reads[i] for i in sample if i < 24000], "two fill streams, config 2")
    at = 24000
    for gi, (seqs, edges, rs) in enumerate(fuzzgen.cases(909, 300, 10)):
        compare(out[2][0][at:at + len(rs)], checker.align_batch(seqs, edges, rs), rs, "two fill streams, fuzz graph %d" % gi)
        at += len(rs)
