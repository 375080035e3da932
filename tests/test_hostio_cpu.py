"""CPU checks of the host input layer (JSON / FASTA / BAM+BAI / read extraction / graph descriptions / manifests /
statistics): the C++ program tests/host_cpp/test_hostio holds the expectations of the reference's unit tests; here it is
built and run, and the BAM reader is additionally compared record by record with an independent decoder (Python gzip +
struct on the same file)."""
import gzip
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SITES = os.path.join(ROOT, "tests", "golden", "sites")


@pytest.fixture(scope="module")
def hostio():
    from paragraph_amd import build
    build.build_genotyping_test()
    exe = os.path.join(ROOT, "tests", "host_cpp", "test_hostio")
    assert os.path.exists(exe)
    return exe


def test_hostio_program(hostio):
    r = subprocess.run([hostio, SITES], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL OK" in r.stdout


def decode_bam(path):
    """SAM/BAM spec section 4.2, straight from the inflated stream (BGZF = concatenated gzip members)."""
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", data, 4)
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, at)
    at += 4
    names = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, at)
        names.append(data[at + 4:at + 4 + l_name - 1].decode())
        at += 4 + l_name + 4
    recs = []
    while at < len(data):
        block_size, = struct.unpack_from("<i", data, at)
        tid, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, mtid, mpos, _tlen = struct.unpack_from("<iiBBHHHiiii", data, at + 4)
        p = at + 36
        name = data[p:p + l_read_name - 1].decode()
        p += l_read_name
        cigar = struct.unpack_from("<%dI" % n_cigar, data, p)
        p += 4 * n_cigar
        seq = "".join("=ACMGRSVTWYHKDBN"[(data[p + i // 2] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq))
        p += (l_seq + 1) // 2
        qual = "".join(chr(33 + q) for q in data[p:p + l_seq])
        ref_span = sum(c >> 4 for c in cigar if (c & 15) in (0, 2, 3, 7, 8))
        if flag & 4 or not cigar or ref_span == 0:
            ref_span = 1
        recs.append(dict(name=name, tid=tid, pos=pos, mapq=mapq, flag=flag, mtid=mtid, mpos=mpos, seq=seq, qual=qual, end=pos + ref_span))
        at += 4 + block_size
    return names, recs


def dump(hostio, bam, region):
    r = subprocess.run([hostio, "--dump-bam", bam, region], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return [line.split("\t") for line in r.stdout.splitlines()]


@pytest.mark.parametrize("region", ["chrX", "chrX:1-1000", "chrX:900-1100", "chrX:8,500-9,000", "chrX:1149", "chrX:5000-6000", "chr1"])
def test_bam_reader_matches_independent_decoder(hostio, region):
    bam = os.path.join(SITES, "chrX", "chrX_graph_typing.bam")
    names, recs = decode_bam(bam)
    chrom, _, rng = region.partition(":")
    tid = names.index(chrom)
    beg, end = 0, 1 << 29
    if rng:
        lo, _, hi = rng.replace(",", "").partition("-")
        beg = int(lo) - 1
        end = int(hi) if hi else 1 << 29
    want = [r for r in recs if r["tid"] == tid and r["pos"] < end and r["end"] > beg and not r["flag"] & 0x900]
    got = dump(hostio, bam, region)
    assert len(got) == len(want)
    if region == "chrX":
        assert len(got) > 400
    for g, w in zip(got, want):
        flags = "%d%d%d%d%d" % (not w["flag"] & 4, bool(w["flag"] & 0x40), not w["flag"] & 8, bool(w["flag"] & 0x10), bool(w["flag"] & 0x20))
        assert g == [w["name"], str(w["tid"]), str(w["pos"]), str(w["mapq"]), flags, str(w["mtid"]), str(w["mpos"]), w["seq"], w["qual"]]


def test_bam_reader_on_synthetic_multiblock_bam(hostio, tmp_path):
    """Thousands of records over several BGZF blocks, two contigs, soft clips / deletions / unmapped mates placed with their
    partner, secondary records to be skipped: region queries through the written BAI equal the filter over all records."""
    import random
    from tests import bamwriter
    rng = random.Random(12)
    recs = []
    for tid, length in ((0, 300000), (1, 90000)):
        pos = 0
        while True:
            pos += rng.choice([0, 1, 3, 40, 200]) if rng.random() < 0.998 else 30000
            if pos > length - 200:
                break
            L = rng.randint(30, 150)
            kind = rng.random()
            cigar = [(L, "M")]
            if kind < 0.1:
                cigar = [(10, "S"), (L - 10, "M")]
            elif kind < 0.2:
                cigar = [(20, "M"), (rng.choice([5, 5000]), "D"), (L - 20, "M")]
            flag = rng.choice([0x63, 0x93, 0x53, 0xa3, 0x100 | 0x63, 0x800 | 0x63, 0x4 | 0x41])
            recs.append(dict(name="r%d_%d" % (tid, len(recs)), tid=tid, pos=pos, seq="".join(rng.choice("ACGTN") for _ in range(L)),
                             qual="".join(chr(33 + rng.randint(0, 41)) for _ in range(L)), flag=flag, mapq=rng.randint(0, 60), mtid=tid,
                             mpos=max(0, pos + rng.randint(-400, 400)), cigar=cigar))
    assert len(recs) > 3000
    bam = str(tmp_path / "synthetic.bam")
    bamwriter.write_bam(bam, [("ctgA", 300000), ("ctgB", 90000)], recs)
    names, decoded = decode_bam(bam)
    assert names == ["ctgA", "ctgB"] and len(decoded) == len(recs)
    for region in ["ctgA", "ctgB", "ctgA:1-20000", "ctgA:16,300-16,500", "ctgA:100000-100001", "ctgB:40000-89999", "ctgA:299000", "ctgB:1-1"] + [
            "ctgA:%d-%d" % (b, b + rng.randint(0, 3000)) for b in (rng.randint(1, 299000) for _ in range(25))]:
        chrom, _, span = region.partition(":")
        tid = names.index(chrom)
        beg, end = 0, 1 << 29
        if span:
            lo, _, hi = span.replace(",", "").partition("-")
            beg, end = int(lo) - 1, int(hi) if hi else 1 << 29
        want = [r for r in decoded if r["tid"] == tid and r["pos"] < end and r["end"] > beg and not r["flag"] & 0x900]
        got = dump(hostio, bam, region)
        assert [g[0] for g in got] == [w["name"] for w in want], region
        for g, w in zip(got, want):
            assert g[2] == str(w["pos"]) and g[7] == w["seq"] and g[8] == w["qual"]


def _reference_bams():
    root = "/root/reference/share/test-data"
    found = []
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".bam") and (os.path.exists(os.path.join(dp, f + ".bai")) or os.path.exists(os.path.join(dp, f[:-4] + ".bai"))):
                found.append(os.path.join(dp, f))
    return sorted(found)


@pytest.mark.parametrize("bam", _reference_bams() or [None])
def test_bam_reader_on_the_reference_test_bams(hostio, bam):
    """Every indexed BAM of the reference's test data (samtools-written: records span BGZF blocks, many contigs, real
    indexes), read through the index contig by contig and in windows, against the independent decoder.  Runs only where
    /root/reference is mounted (the build container)."""
    if bam is None:
        pytest.skip("/root/reference is not mounted here")
    names, recs = decode_bam(bam)
    primary = [r for r in recs if not r["flag"] & 0x900]
    by_tid = {}
    for r in primary:
        by_tid.setdefault(r["tid"], []).append(r)
    checked = 0
    for tid, rs in sorted(by_tid.items()):
        if tid < 0:
            continue
        got = dump(hostio, bam, names[tid])
        assert [(g[0], g[2], g[7]) for g in got] == [(w["name"], str(w["pos"]), w["seq"]) for w in rs], (bam, names[tid])
        checked += len(got)
        # a few windows inside the covered span
        lo, hi = rs[0]["pos"], rs[-1]["pos"]
        for k in range(4):
            beg = lo + (hi - lo) * k // 4
            end = beg + 700
            want = [w for w in rs if w["pos"] < end and w["end"] > beg]
            got = dump(hostio, bam, "%s:%d-%d" % (names[tid], beg + 1, end))
            assert [(g[0], g[2]) for g in got] == [(w["name"], str(w["pos"])) for w in want], (bam, names[tid], beg, end)
    assert checked == len([r for r in primary if r["tid"] >= 0])


def test_damaged_inputs_are_refused_cleanly(hostio, tmp_path):
    """Bit flips and truncation in a BAM, its index and a graph description end in an error message (exit 1) or in a normal
    run (exit 0) -- never in a crash or a runaway allocation.  (The same loop was run under ASan + UBSan, 1 100 variants.)"""
    import random
    rng = random.Random(3)
    src = os.path.join(SITES, "chrX", "chrX_graph_typing.bam")
    bam, bai = open(src, "rb").read(), open(src + ".bai", "rb").read()
    x = str(tmp_path / "x.bam")
    for it in range(120):
        b, i = bytearray(bam), bytearray(bai)
        kind = rng.choice(["bam", "bai", "both", "trunc", "truncbai"])
        if kind in ("bam", "both"):
            for _ in range(rng.randint(1, 6)):
                b[rng.randrange(len(b))] = rng.randrange(256)
        if kind in ("bai", "both"):
            for _ in range(rng.randint(1, 6)):
                i[rng.randrange(len(i))] = rng.randrange(256)
        if kind == "trunc":
            b = b[:rng.randrange(20, len(b))]
        if kind == "truncbai":
            i = i[:rng.randrange(4, len(i))]
        open(x, "wb").write(b)
        open(x + ".bai", "wb").write(i)
        r = subprocess.run([hostio, "--dump-bam", x, rng.choice(["chrX", "chrX:800-1200", "chrX:8000-9000", "chr1"])],
                           capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 1), (it, kind, r.returncode, r.stderr[-300:])
        assert r.returncode == 0 or r.stderr.startswith("error: "), (it, kind, r.stderr[-300:])
    text = open(os.path.join(SITES, "chrX", "chrX_graph_typing.2sample.json")).read()
    fasta = os.path.join(SITES, "chrX", "chrX_graph_typing.fa")
    g = str(tmp_path / "g.json")
    for it in range(120):
        t = list(text)
        for _ in range(rng.randint(1, 5)):
            k = rng.randrange(len(t))
            t[k] = rng.choice('{}[]",:0123456789abcXN-\\ \n')
        open(g, "w").write("".join(t))
        r = subprocess.run([hostio, "--load-graph", g, fasta], capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 1), (it, r.returncode, r.stderr[-300:])


def test_synth_e2e_dataset_reads_back(hostio, tmp_path):
    """The benchmark's data-set maker (paragraph_amd/synth_e2e.py: vectorised records, BGZF blocks deflated by forked workers,
    virtual offsets assigned at the join): the BAM decodes to exactly the records that were drawn (independent decoder), and
    the product's reader answers region queries through the written BAI with the filter over all records."""
    import numpy as np
    from paragraph_amd import synth_e2e
    keep = [0, 7, 33, 59]
    d = synth_e2e.make_dataset(str(tmp_path / "e2e"), n_sites=60, depth=30.0, seed=5, procs=3, keep_sites=keep)
    names, decoded = decode_bam(d["bam"])
    assert names == ["chr1"] and len(decoded) == d["reads"] > 60 * 200
    assert all(a["pos"] <= b["pos"] for a, b in zip(decoded, decoded[1:]))
    by_name = {}
    for r in decoded:
        by_name.setdefault(r["name"], []).append(r)
    assert all(len(v) == 2 for v in by_name.values())
    for i in keep:
        k = d["kept"][i]
        for row in range(len(k["pos"])):
            name = "s%06d_f%04d" % (i, k["fragment"][row])
            mine = [r for r in by_name[name] if r["flag"] == int(k["flag"][row])]
            assert len(mine) == 1 and mine[0]["pos"] == int(k["pos"][row]) and mine[0]["mpos"] == int(k["mpos"][row])
            assert mine[0]["seq"] == k["bases"][row].tobytes().decode() and mine[0]["qual"] == "I" * 150
    glen = synth_e2e.SPACING * 61
    for region in ["chr1", "chr1:1-3000", "chr1:2851-3400", "chr1:%d-%d" % (glen - 5000, glen), "chr1:16384-16385", "chr1:90000-100000"]:
        chrom, _, span = region.partition(":")
        beg, end = 0, 1 << 29
        if span:
            lo, _, hi = span.partition("-")
            beg, end = int(lo) - 1, int(hi)
        want = [r for r in decoded if r["pos"] < end and r["end"] > beg]
        got = dump(hostio, d["bam"], region)
        assert [g[0] for g in got] == [w["name"] for w in want], region
        assert all(g[2] == str(w["pos"]) and g[7] == w["seq"] for g, w in zip(got, want))
    # the graph descriptions and the truth are the ones tools/e2e/make_sites.py would write for the same draw
    import json
    g0 = json.load(open(d["graphs"][0]))
    assert g0 == d["sites"][0].graph() and g0["ID"] == "site_0" and d["truth"][0]["ID"] == "site_0"
    assert open(d["reference"] + ".fai").read().split("\t")[1] == str(glen)
    ref_text = "".join(l.strip() for l in open(d["reference"]) if not l.startswith(">"))
    assert ref_text == np.asarray(d["ref"]).tobytes().decode()
