"""Synthetic many-site input for the BAM -> genotypes leg of bench.py and the workflow tests: a random reference, N deletion /
insertion / swap sites spaced along it (graph descriptions with reference-interval nodes, share/schema/graph_schema.json),
ONE coordinate-sorted BAM (+ .bai) of paired reads sampled around every site from a diploid genome that carries each alternate
allele with genotype 0/0, 0/1 or 1/1, a grmpy manifest and the simulated truth.

The same data set tools/e2e/make_sites.py writes record by record, made fast enough for the inside of a benchmark run: reads
are drawn with numpy per site, BAM records are fixed-size rows of one byte matrix (equal name and read lengths), and the BGZF
blocks are deflated by forked workers (blocks are independent; virtual offsets are assigned when the pieces are joined).
10 000 sites at 30x = 2.8 M reads: ~6 s on 16 cores.  A data-set maker, not part of the realignment path: the product
only READS BAMs (host/src/io.cpp).

    python -m paragraph_amd.synth_e2e <outdir> [n_sites] [depth] [seed] [procs]
"""
import json
import os
import struct
import sys
import zlib

import numpy as np

SPACING, FLANK, FRAG_MEAN, FRAG_SD, REACH = 3000, 150, 400, 40, 700
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_NIB = np.zeros(256, dtype=np.uint8)
for _c, _v in zip(b"=ACMGRSVTWYHKDBN", range(16)):
    _NIB[_c] = _v
BGZF_BLOCK = 0xff00


def _record_dtype(read_len, name_len):
    # SAM/BAM specification v1 section 4.2: block_size, then the 32-byte core, name, cigar, 4-bit bases, qualities
    return np.dtype([("block_size", "<i4"), ("refID", "<i4"), ("pos", "<i4"), ("l_read_name", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                     ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_refID", "<i4"), ("next_pos", "<i4"), ("tlen", "<i4"),
                     ("name", "u1", (name_len,)), ("cigar", "<u4"), ("seq", "u1", ((read_len + 1) // 2,)), ("qual", "u1", (read_len,))])


def _reg2bin(beg, end):
    """UCSC binning (SAM specification section 5.3), vectorised; end exclusive"""
    end = end - 1
    out = np.zeros(len(beg), dtype=np.int64)
    done = np.zeros(len(beg), dtype=bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((beg >> shift) == (end >> shift))
        out[hit] = base + (beg[hit] >> shift)
        done |= hit
    return out


def _bgzf_block(payload, level):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = comp.compress(payload) + comp.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(cdata) + 25) + cdata
            + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


class SiteSpec:
    """one site: where it is, what it replaces with what, the genotype drawn for the sample"""
    __slots__ = ("index", "start", "del_len", "ins", "gt", "kind")

    def __init__(self, index, start, del_len, ins, gt, kind):
        self.index, self.start, self.del_len, self.ins, self.gt, self.kind = index, start, del_len, ins, gt, kind

    @property
    def end(self):
        return self.start + self.del_len

    def graph(self):
        """the graph description make_sites.py writes: source, LF, [REF], [INS], RF, sink; REF / ALT paths and labels"""
        start, end = self.start, self.end
        lf, rf = (start - FLANK, start), (end, end + FLANK)
        nodes = [{"name": "source", "sequence": "NNNNNNNNNN"}, {"name": "LF", "reference": "chr1:%d-%d" % (lf[0] + 1, lf[1])}]
        edges = [{"from": "source", "to": "LF"}]
        if self.del_len:
            nodes.append({"name": "REF", "reference": "chr1:%d-%d" % (start + 1, end)})
            edges.append({"from": "LF", "to": "REF", "sequences": ["REF"]})
        if self.ins:
            nodes.append({"name": "INS", "sequence": self.ins})
            edges.append({"from": "LF", "to": "INS", "sequences": ["ALT"]})
        nodes += [{"name": "RF", "reference": "chr1:%d-%d" % (rf[0] + 1, rf[1])}, {"name": "sink", "sequence": "NNNNNNNNNN"}]
        if self.del_len:
            edges.append({"from": "REF", "to": "RF", "sequences": ["REF"]})
        if self.ins:
            edges.append({"from": "INS", "to": "RF", "sequences": ["ALT"]})
        if not self.del_len:
            edges.append({"from": "LF", "to": "RF", "sequences": ["REF"]})
        if not self.ins:
            edges.append({"from": "LF", "to": "RF", "sequences": ["ALT"]})
        edges.append({"from": "RF", "to": "sink"})
        order = {n["name"]: k for k, n in enumerate(nodes)}
        edges.sort(key=lambda e: (order[e["from"]], order[e["to"]]))
        mid_ref = ["REF"] if self.del_len else []
        mid_alt = ["INS"] if self.ins else []
        return {"ID": "site_%d" % self.index, "nodes": nodes, "edges": edges, "sequencenames": ["ALT", "REF"],
                "target_regions": ["chr1:%d-%d" % (lf[0] + 1, rf[1])],
                "paths": [{"nodes": ["source", "LF"] + mid_ref + ["RF", "sink"], "path_id": "REF|1", "sequence": "REF"},
                          {"nodes": ["source", "LF"] + mid_alt + ["RF", "sink"], "path_id": "ALT|1", "sequence": "ALT"}]}

    def truth(self):
        return {"ID": "site_%d" % self.index, "gt": "/".join("ALT" if a else "REF" for a in sorted(self.gt, reverse=True)), "kind": self.kind}


def draw_sites(n_sites, seed):
    rng = np.random.default_rng([seed, 0x51735])
    sites = []
    for i in range(n_sites):
        kind = ("del", "ins", "swap")[int(rng.integers(3))]
        del_len = 0 if kind == "ins" else int(rng.integers(20, 301))
        ins = "" if kind == "del" else _ACGT[rng.integers(0, 4, int(rng.integers(10, 121)))].tobytes().decode()
        gt = ((0, 0), (0, 1), (1, 1))[int(rng.integers(3))]
        sites.append(SiteSpec(i, SPACING * (i + 1), del_len, ins, gt, kind))
    return sites


def site_reads(ref, site, depth, read_len, rng, sub_rate=0.0023):
    """Paired reads of one site from its two haplotypes over [start - 700, end + 700): (pos, mate pos, bases (n, L), flag,
    fragment number), two rows per fragment (first mate forward, second mate reverse; bases as a BAM stores them: forward)."""
    start, end = site.start, site.end
    lo, hi = start - REACH, end + REACH
    ins = np.frombuffer(site.ins.encode(), dtype=np.uint8)
    haps = [np.concatenate([ref[lo:start], ins if a else ref[start:end], ref[end:hi]]) for a in site.gt]
    n_frag = int(depth * (hi - lo) / (2 * read_len))
    h = rng.integers(0, 2, n_frag)
    fl = np.maximum(read_len + 10, rng.normal(FRAG_MEAN, FRAG_SD, n_frag).astype(np.int64))
    hap_len = np.array([len(haps[0]), len(haps[1])])[h]
    a = (rng.random(n_frag) * (hap_len - fl)).astype(np.int64)
    alt = np.array(site.gt)[h].astype(bool)
    cols = np.arange(read_len)
    bases = np.empty((2 * n_frag, read_len), dtype=np.uint8)
    for k in (0, 1):
        sel = h == k
        off1 = a[sel]
        off2 = a[sel] + fl[sel] - read_len
        rows = np.nonzero(sel)[0]
        bases[2 * rows] = haps[k][off1[:, None] + cols]
        bases[2 * rows + 1] = haps[k][off2[:, None] + cols]

    def to_ref(x):  # haplotype offset -> approximate linear position (what a mapper would report)
        plain = lo + x
        shifted = np.maximum(start, lo + x - len(ins) + site.del_len)
        return np.where((x <= start - lo) | ~alt, plain, shifted)
    p1, p2 = to_ref(a), to_ref(a + fl - read_len)
    errs = rng.random(bases.shape) < sub_rate
    bases[errs] = _ACGT[rng.integers(0, 4, int(errs.sum()))]
    pos = np.empty(2 * n_frag, dtype=np.int64)
    mpos = np.empty(2 * n_frag, dtype=np.int64)
    pos[0::2], pos[1::2], mpos[0::2], mpos[1::2] = p1, p2, p2, p1
    flag = np.empty(2 * n_frag, dtype=np.uint16)
    flag[0::2], flag[1::2] = 0x63, 0x93
    return pos, mpos, bases, flag, np.repeat(np.arange(n_frag), 2)


def _names(site_index, frag, name_len):
    """"s<site:06>_f<frag:04>" + NUL as rows of bytes"""
    out = np.zeros((len(frag), name_len), dtype=np.uint8)
    out[:, 0] = ord("s")
    for j in range(6):
        out[:, 1 + j] = 48 + (site_index // 10 ** (5 - j)) % 10
    out[:, 7], out[:, 8] = ord("_"), ord("f")
    for j in range(4):
        out[:, 9 + j] = 48 + (frag // 10 ** (3 - j)) % 10
    return out


NAME_LEN = 14


_SHARED = {}  # the reference array, set before the workers are forked (30 MB that must not travel with every job)


def _worker(job):
    """records of a run of sites: BGZF blocks (bytes), their compressed sizes, and per record (pos, block, offset in block)"""
    sites, depth, read_len, seed, level, keep, graph_dir = job
    ref = _SHARED["ref"]
    dt = _record_dtype(read_len, NAME_LEN)
    parts, kept = [], {}
    for site in sites:
        rng = np.random.default_rng([seed, 0xbead5, site.index])
        pos, mpos, bases, flag, frag = site_reads(ref, site, depth, read_len, rng)
        rec = np.zeros(len(pos), dtype=dt)
        rec["block_size"] = dt.itemsize - 4
        rec["pos"], rec["next_pos"] = pos, mpos
        rec["l_read_name"], rec["mapq"], rec["n_cigar"], rec["l_seq"] = NAME_LEN, 60, 1, read_len
        rec["bin"] = _reg2bin(pos, pos + read_len)
        rec["flag"] = flag
        rec["name"] = _names(site.index, frag, NAME_LEN)
        rec["cigar"] = read_len << 4
        nib = _NIB[bases]
        if read_len & 1:
            nib = np.concatenate([nib, np.zeros((len(nib), 1), np.uint8)], axis=1)
        rec["seq"] = (nib[:, 0::2] << 4) | nib[:, 1::2]
        rec["qual"] = 40
        parts.append(rec)
        if site.index in keep:
            kept[site.index] = {"pos": pos, "mpos": mpos, "bases": bases, "flag": flag, "fragment": frag}
        if graph_dir:
            with open(os.path.join(graph_dir, "site_%d.json" % site.index), "w") as f:
                json.dump(site.graph(), f)
    rec = np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
    rec = rec[np.argsort(rec["pos"], kind="stable")]
    per_block = BGZF_BLOCK // dt.itemsize
    raw = rec.view(np.uint8).reshape(len(rec), dt.itemsize)
    blocks = [_bgzf_block(raw[b:b + per_block].tobytes(), level) for b in range(0, len(rec), per_block)]
    return (b"".join(blocks), np.array([len(b) for b in blocks], dtype=np.int64), rec["pos"].astype(np.int64), per_block, dt.itemsize,
            kept)


def write_fasta(path, name, ref):
    n = len(ref)
    full = n // 60 * 60
    body = np.empty((full // 60, 61), dtype=np.uint8)
    body[:, :60] = ref[:full].reshape(-1, 60)
    body[:, 60] = 10
    head = (">%s\n" % name).encode()
    with open(path, "wb") as f:
        f.write(head)
        f.write(body.tobytes())
        if full < n:
            f.write(ref[full:].tobytes() + b"\n")
    with open(path + ".fai", "w") as f:
        f.write("%s\t%d\t%d\t60\t61\n" % (name, n, len(head)))


def _write_bai(path, n_ref_windows, bins, beg_v, end_v, pos, read_len):
    """BAI (SAM specification section 5.2) of one contig from per-record bins and virtual offsets, records in file order"""
    order = np.argsort(bins, kind="stable")
    b_sorted = bins[order]
    cuts = np.nonzero(np.diff(b_sorted))[0] + 1
    groups = np.split(order, cuts)
    out = bytearray(b"BAI\1" + struct.pack("<i", 1) + struct.pack("<i", len(groups) if len(bins) else 0))
    for g in groups if len(bins) else []:
        b, e = beg_v[g], end_v[g]
        new = np.ones(len(g), dtype=bool)
        new[1:] = b[1:] > e[:-1]  # a record that starts where the previous one of its bin ended extends that chunk
        firsts = np.nonzero(new)[0]
        lasts = np.append(firsts[1:] - 1, len(g) - 1)
        out += struct.pack("<Ii", int(bins[g[0]]), len(firsts))
        for f, l in zip(firsts, lasts):
            out += struct.pack("<QQ", int(b[f]), int(e[l]))
    lin = np.full(n_ref_windows, np.iinfo(np.int64).max, dtype=np.int64)
    for w in (pos >> 14, (pos + read_len - 1) >> 14):
        np.minimum.at(lin, w, beg_v)
    n_intv = int(((pos + read_len - 1) >> 14).max()) + 1 if len(pos) else 0
    lin = lin[:n_intv]
    prev = 0
    vals = []
    for v in lin.tolist():
        prev = v if v != np.iinfo(np.int64).max else prev
        vals.append(prev)
    out += struct.pack("<i", n_intv) + struct.pack("<%dQ" % n_intv, *vals)
    with open(path, "wb") as f:
        f.write(out)


def _reference(n_sites, seed):
    rng = np.random.default_rng([seed, 0x9e0])
    return _ACGT[rng.integers(0, 4, SPACING * (n_sites + 1))]


def make_part(outdir, part, n_parts, n_sites=10000, depth=30.0, seed=1, read_len=150, procs=1, level=6, keep_sites=()):
    """The BGZF pieces (and graph files) of sites [n_sites * part / n_parts, n_sites * (part + 1) / n_parts) -> outdir/part_<part>.pkl.
    Parts are independent, so the ranks of a multi-process benchmark each make one; `join` assembles the data set."""
    import pickle
    gdir = os.path.join(outdir, "graphs")
    os.makedirs(gdir, exist_ok=True)
    ref = _reference(n_sites, seed)
    sites = draw_sites(n_sites, seed)[n_sites * part // n_parts:n_sites * (part + 1) // n_parts]
    keep = frozenset(int(i) for i in keep_sites)
    procs = max(1, min(procs, max(1, len(sites))))
    per = max(1, min(256, (len(sites) + 4 * procs - 1) // (4 * procs)))  # a few jobs per worker: sites differ in depth of work
    jobs = [(sites[i:i + per], depth, read_len, seed, level, keep, gdir) for i in range(0, len(sites), per)]
    _SHARED["ref"] = ref
    if procs == 1:
        pieces = [_worker(j) for j in jobs]
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            pieces = pool.map(_worker, jobs, chunksize=1)
    path = os.path.join(outdir, "part_%d.pkl" % part)
    with open(path + ".tmp", "wb") as f:
        pickle.dump(pieces, f, protocol=4)
    os.replace(path + ".tmp", path)  # complete or absent: `join` may be polling for it from another process
    return path


def join(outdir, n_parts, n_sites=10000, depth=30.0, seed=1, read_len=150, level=6, wait_s=0.0):
    """Assembles ref.fa(.fai), reads.bam(.bai), graphs.txt, manifest.txt, truth.json from the parts (waiting up to wait_s for
    parts other processes are still writing) and returns the description make_dataset documents."""
    import pickle
    import time
    glen = SPACING * (n_sites + 1)
    ref = _reference(n_sites, seed)
    write_fasta(os.path.join(outdir, "ref.fa"), "chr1", ref)
    sites = draw_sites(n_sites, seed)
    # header block, the parts' blocks in site order, EOF marker; virtual offsets from the running compressed size
    text = ("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % glen).encode()
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", glen)
    bam_path = os.path.join(outdir, "reads.bam")
    at = 0
    beg_v, end_v, all_pos, kept = [], [], [], {}
    deadline = time.time() + wait_s
    with open(bam_path, "wb") as f:
        hb = _bgzf_block(head, level)
        f.write(hb)
        at += len(hb)
        for part in range(n_parts):
            path = os.path.join(outdir, "part_%d.pkl" % part)
            while not os.path.exists(path):
                if time.time() > deadline:
                    raise RuntimeError("synth_e2e.join: %s did not appear" % path)
                time.sleep(0.05)
            with open(path, "rb") as pf:
                pieces = pickle.load(pf)
            os.unlink(path)
            for data, sizes, pos, per_block, item, k in pieces:
                kept.update(k)
                f.write(data)
                starts = at + np.concatenate([[0], np.cumsum(sizes)[:-1]]) if len(sizes) else np.zeros(0, np.int64)
                idx = np.arange(len(pos))
                blk, within = idx // per_block, idx % per_block
                b = (starts[blk].astype(np.int64) << 16) | (within * item)
                beg_v.append(b)
                end_v.append(b + item)  # "offset = the block's length" is a valid virtual offset of the end of its last record
                all_pos.append(pos)
                at += int(sizes.sum())
        f.write(_bgzf_block(b"", level))
    beg_v = np.concatenate(beg_v) if beg_v else np.zeros(0, np.int64)
    end_v = np.concatenate(end_v) if end_v else np.zeros(0, np.int64)
    all_pos = np.concatenate(all_pos) if all_pos else np.zeros(0, np.int64)
    assert (np.diff(all_pos) >= 0).all(), "the pieces must join into a coordinate-sorted file (sites are 3 kb apart)"
    _write_bai(bam_path + ".bai", (glen >> 14) + 2, _reg2bin(all_pos, all_pos + read_len), beg_v, end_v, all_pos, read_len)
    graphs = [os.path.join(outdir, "graphs", "site_%d.json" % i) for i in range(n_sites)]
    with open(os.path.join(outdir, "graphs.txt"), "w") as f:
        f.write("\n".join(graphs) + "\n")
    manifest = os.path.join(outdir, "manifest.txt")
    with open(manifest, "w") as f:
        f.write("id\tpath\tdepth\tread length\nSYN\t%s\t%g\t%d\n" % (bam_path, depth, read_len))
    truth = [s.truth() for s in sites]
    with open(os.path.join(outdir, "truth.json"), "w") as f:
        json.dump(truth, f)
    return {"reference": os.path.join(outdir, "ref.fa"), "manifest": manifest, "graphs": graphs, "truth": truth, "sites": sites,
            "reads": int(len(all_pos)), "bam": bam_path, "bam_bytes": os.path.getsize(bam_path), "kept": kept, "ref": ref}


def make_dataset(outdir, n_sites=10000, depth=30.0, seed=1, read_len=150, procs=1, level=6, keep_sites=()):
    """Writes ref.fa(.fai), reads.bam(.bai), graphs/site_<i>.json, graphs.txt, manifest.txt, truth.json under outdir.
    Returns {"reference", "manifest", "graphs": [paths], "truth": [...], "sites": [SiteSpec], "reads": n, "bam", "bam_bytes",
             "kept": {site index: arrays of its reads}, "ref": the reference as a uint8 array} (kept: the reads of `keep_sites`,
    for a checker)."""
    make_part(outdir, 0, 1, n_sites, depth, seed, read_len, procs, level, keep_sites)
    return join(outdir, 1, n_sites, depth, seed, read_len, level)


def loaded_graph(site, ref):
    """The graph as grm::graphFromJson loads the description (src/c++/lib/grm/GraphInput.cpp:60-140): source / sink become one-base
    "X" nodes, reference nodes take their interval.  -> (names, sequences, [(from, to)], {(from, to): [labels]})"""
    spec = site.graph()
    names = [n["name"] for n in spec["nodes"]]
    seqs = []
    for k, n in enumerate(spec["nodes"]):
        if k in (0, len(names) - 1) and n["name"].upper() in ("SOURCE", "SINK"):
            seqs.append("X")
        elif "sequence" in n:
            seqs.append(n["sequence"])
        else:
            a, b = n["reference"].split(":")[1].split("-")
            seqs.append(np.asarray(ref[int(a) - 1:int(b)]).tobytes().decode())
    index = {n: k for k, n in enumerate(names)}
    edges = [(index[e["from"]], index[e["to"]]) for e in spec["edges"]]
    labels = {(index[e["from"]], index[e["to"]]): list(e["sequences"]) for e in spec["edges"] if e.get("sequences")}
    return names, seqs, edges, labels


def extracted(site, reads, read_len=150):
    """Row mask of the reads common::extractReads keeps for the site's one target region (src/c++/lib/common/ReadExtraction.cpp:
    135-178: a read is kept when it or its mate -- placed with the read's own length -- touches the 0-based inclusive region;
    nothing here is far enough from its mate for recoverMissingMates, and no site reaches max_reads)."""
    rs, re_ = site.start - FLANK, site.end + FLANK - 1
    own = ~((reads["pos"] > re_) | (reads["pos"] + read_len < rs))
    mate = ~((reads["mpos"] > re_) | (reads["mpos"] + read_len < rs))
    return own | mate


if __name__ == "__main__":
    a = sys.argv
    d = make_dataset(a[1], int(a[2]) if len(a) > 2 else 2000, float(a[3]) if len(a) > 3 else 30.0, int(a[4]) if len(a) > 4 else 1,
                     procs=int(a[5]) if len(a) > 5 else (os.cpu_count() or 1))
    print("wrote %d sites, %d reads, %d-byte BAM to %s" % (len(d["graphs"]), d["reads"], d["bam_bytes"], a[1]))
