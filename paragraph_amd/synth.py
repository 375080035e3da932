"""Deterministic synthetic sites and reads for the workloads named in BASELINE.json / SURVEY.md 8(d).

Everything is driven by splitmix64 so that a (seed, parameters) pair names a data set exactly; no
real genomes or BAMs are needed.  Graph shapes follow the reference's own templates:

* ``del_site``     src/python/lib/grm/graph_templates/shortdeletion.py:29-95   (LF, MID, RF)
* ``ins_site``     src/python/lib/grm/graph_templates/insertion.py:29-95       (LF, INS, RF)
* ``longdel_site`` src/python/lib/grm/graph_templates/longdeletion.py:28-131   (6 nodes, 2 regions)
"""
import numpy as np

_MASK = (1 << 64) - 1
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


class SplitMix64:
    """Scalar + vector splitmix64 (Steele/Lea/Flood) stream."""

    def __init__(self, seed):
        self.state = seed & _MASK

    def next(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & _MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n

    def uniform(self):
        return (self.next() >> 11) * (1.0 / (1 << 53))

    def vector(self, n):
        """n outputs as uint64 (identical to calling next() n times)."""
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = np.uint64(self.state) + idx * np.uint64(0x9E3779B97F4A7C15)
            self.state = int(z[-1]) if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform_vector(self, n):
        return (self.vector(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def random_contig(seed, n):
    """Uniform ACGT contig of length n as bytes."""
    rng = SplitMix64(seed)
    return _ACGT[(rng.vector(n) & np.uint64(3)).astype(np.int64)].tobytes()


def revcomp(b):
    return _COMP[np.frombuffer(b, dtype=np.uint8)][::-1].tobytes()


class Site:
    """One variant graph: node sequences (bytes), edges (from,to), names, edge labels, haplotypes."""

    def __init__(self, kind, names, seqs, edges, labels, haplotypes):
        self.kind = kind
        self.names = names
        self.seqs = [s.decode() if isinstance(s, bytes) else s for s in seqs]
        self.edges = edges
        self.labels = labels          # {(from,to): [sequence labels]}
        self.haplotypes = haplotypes  # {label: [node ids]}

    @property
    def total_len(self):
        return sum(len(s) for s in self.seqs)

    def haplotype_seq(self, label):
        return "".join(self.seqs[i] for i in self.haplotypes[label])

    def to_json(self):
        """Graph JSON in the reference's input schema (share/schema/input_schema.json), inline sequences."""
        return {
            "sequencenames": sorted(self.haplotypes),
            "nodes": [{"name": n, "sequence": s} for n, s in zip(self.names, self.seqs)],
            "edges": [{"from": self.names[f], "to": self.names[t], "sequences": self.labels.get((f, t), [])}
                      for f, t in self.edges],
            "paths": [{"nodes": [self.names[i] for i in nodes], "path_id": "%s|1" % lab, "sequence": lab}
                      for lab, nodes in sorted(self.haplotypes.items())],
        }


def del_site(contig, start, end, flank=150):
    """1-based inclusive deleted interval [start, end] on `contig` (bytes)."""
    lf = contig[max(1, start - flank - 1) - 1:start - 1]
    mid = contig[start - 1:end]
    rf = contig[end:end + flank + 1]
    return Site("del", ["LF", "MID", "RF"], [lf, mid, rf], [(0, 1), (0, 2), (1, 2)],
                {(0, 2): ["DEL"], (0, 1): ["REF"], (1, 2): ["REF"]}, {"REF": [0, 1, 2], "DEL": [0, 2]})


def ins_site(contig, start, ins, flank=150):
    """Insertion of `ins` after 1-based position `start`."""
    lf = contig[max(1, start - flank - 1) - 1:max(1, start - 1)]
    rf = contig[start:start + flank + 1]
    return Site("ins", ["LF", "INS", "RF"], [lf, ins, rf], [(0, 1), (0, 2), (1, 2)],
                {(0, 2): ["REF"], (0, 1): ["INS"], (1, 2): ["INS"]}, {"REF": [0, 2], "INS": [0, 1, 2]})


def longdel_site(contig, start, end, flank=150):
    """Long deletion (deleted length >= 2*flank): six nodes source, LF, MID_L, MID_R, RF, sink with the
    middle of the deleted interval left out (two target regions).  Coordinates, node order and the
    seven edges mirror the reference template, including its MID_R = [end-flank, end-1] interval;
    source/sink get the sequence "X" as grm::graphFromJson assigns (GraphInput.cpp:80-89)."""
    assert end - start + 1 >= 2 * flank
    lf = contig[max(1, start - flank - 1) - 1:max(1, start - 1)]
    mid_l = contig[start - 1:start + flank - 1]
    mid_r = contig[max(1, end - flank) - 1:max(1, end - 1)]
    rf = contig[end:end + flank + 1]
    return Site("longdel", ["source", "LF", "MID_L", "MID_R", "RF", "sink"], [b"X", lf, mid_l, mid_r, rf, b"X"],
                [(0, 1), (0, 3), (1, 4), (1, 2), (3, 4), (3, 5), (4, 5)],
                {(1, 4): ["DEL"], (1, 2): ["REF"], (3, 4): ["REF"]},
                {"DEL": [1, 4], "REF_L": [1, 2], "REF_R": [3, 4]})


def simulate_reads_packed(site, n, read_len, seed, sub_rate=0.01, indel_frac=0.001, random_frac=0.005, n_frac=0.0,
                          hap_weights=None):
    """n reads of length read_len sampled from the site's haplotypes, as an (n, read_len) uint8 array.

    haplotype ~ hap_weights (uniform default), start uniform such that the read fits, strand 50/50,
    i.i.d. substitutions at sub_rate, indel_frac of reads carry one 1-3 bp indel, random_frac are
    unrelated random sequence, n_frac of bases become 'N'."""
    rng = SplitMix64(seed)
    labels = sorted(site.haplotypes)
    haps = [np.frombuffer(site.haplotype_seq(lab).encode(), dtype=np.uint8) for lab in labels]
    haps = [h for h in haps if len(h) >= read_len]
    if not haps:
        raise ValueError("no haplotype long enough for read_len=%d" % read_len)
    w = np.ones(len(haps)) if hap_weights is None else np.asarray(hap_weights, dtype=np.float64)[:len(haps)]
    cdf = np.cumsum(w / w.sum())
    u = rng.uniform_vector(n)
    hap_idx = np.minimum(np.searchsorted(cdf, u, side="right"), len(haps) - 1)
    start_u = rng.uniform_vector(n)
    strand = (rng.vector(n) & np.uint64(1)).astype(bool)
    kind_u = rng.uniform_vector(n)
    out = np.empty((n, read_len), dtype=np.uint8)
    cols = np.arange(read_len)
    for hi, h in enumerate(haps):
        sel = np.nonzero(hap_idx == hi)[0]
        if len(sel) == 0:
            continue
        st = np.floor(start_u[sel] * (len(h) - read_len + 1)).astype(np.int64)
        out[sel] = h[st[:, None] + cols[None, :]]
    # substitutions: replace by one of the three other bases (chunked to bound memory)
    code = np.zeros(256, dtype=np.int64)
    for k, c in enumerate(b"ACGT"):
        code[c] = k
    step = max(1, (1 << 22) // read_len)
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        m = (hi - lo) * read_len
        sub_u = rng.uniform_vector(m).reshape(hi - lo, read_len)
        sub_b = (rng.vector(m) % np.uint64(3)).astype(np.int64).reshape(hi - lo, read_len)
        blk = out[lo:hi]
        out[lo:hi] = np.where(sub_u < sub_rate, _ACGT[(code[blk] + 1 + sub_b) & 3], blk)
    special = np.nonzero(kind_u < random_frac + indel_frac)[0]
    for i in special:
        r = out[i]
        if kind_u[i] < random_frac:
            r = _ACGT[(rng.vector(read_len) & np.uint64(3)).astype(np.int64)]
        else:
            k = 1 + rng.below(3)
            p = 10 + rng.below(max(1, read_len - 20 - k))
            if rng.below(2):  # insertion of k random bases (read keeps its length: tail is dropped)
                ins = _ACGT[(rng.vector(k) & np.uint64(3)).astype(np.int64)]
                r = np.concatenate([r[:p], ins, r[p:]])[:read_len]
            else:  # deletion of k bases (pad the tail with random bases)
                pad = _ACGT[(rng.vector(k) & np.uint64(3)).astype(np.int64)]
                r = np.concatenate([r[:p], r[p + k:], pad])[:read_len]
        out[i] = r
    if n_frac > 0:
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            nu = rng.uniform_vector((hi - lo) * read_len).reshape(hi - lo, read_len)
            out[lo:hi] = np.where(nu < n_frac, np.uint8(ord("N")), out[lo:hi])
    rc = _COMP[out[strand]][:, ::-1]
    out[strand] = rc
    return out


def packed_to_capi(arr):
    """(n, L) uint8 array -> (uint32 offsets[n+1], bytes) as capi.pack_reads returns."""
    n, L = arr.shape
    return (np.arange(n + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32), arr.tobytes()


def simulate_reads(site, n, read_len, seed, **kw):
    """Same data as simulate_reads_packed, as list[str]."""
    arr = simulate_reads_packed(site, n, read_len, seed, **kw)
    return [row.tobytes().decode() for row in arr]


def config2_site(flank=200, del_len=100, contig_seed=1):
    """BASELINE.json configs[1]: 1 DEL graph with 200 bp flanks on a 1 kb synthetic contig (seed 1)."""
    contig = random_contig(contig_seed, 1000)
    start = 400
    return del_site(contig, start, start + del_len - 1, flank=flank)


def config2_reads(n, read_len=150, seed=2):
    site = config2_site()
    return site, simulate_reads(site, n, read_len, seed)


def config2_reads_packed(n, read_len=150, seed=2):
    site = config2_site()
    return site, simulate_reads_packed(site, n, read_len, seed)


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] / [3]: many mixed DEL/INS sites, 30x paired reads
# ---------------------------------------------------------------------------------------------------
def _log_uniform(rng, lo, hi):
    return int(round(float(np.exp(np.log(lo) + rng.uniform() * (np.log(hi) - np.log(lo))))))


class SiteReads:
    """Reads of one site: packed (n, L) uint8 array, fragment ids (mates share one), BAM-strand flags."""

    def __init__(self, site, reads, fragment, is_reverse, genotype):
        self.site = site
        self.reads = reads
        self.fragment = fragment
        self.is_reverse = is_reverse
        self.genotype = genotype


def _paired_reads(rng, hap, keep_lo, keep_hi, depth, read_len, sub_rate, frag0):
    """Paired reads at `depth` over haplotype `hap` (uint8 array); a fragment is kept when at least one mate
    overlaps [keep_lo, keep_hi) (the target region on this haplotype); mate 1 forward, mate 2 reverse."""
    n_frag = int(depth * len(hap) / (2.0 * read_len))
    if n_frag <= 0 or len(hap) < read_len + 2:
        return np.zeros((0, read_len), np.uint8), np.zeros(0, np.uint32), np.zeros(0, np.uint8)
    u = rng.uniform_vector(2 * n_frag)
    # insert ~ N(400, 50) via the sum of 12 uniforms, clipped to [read_len, len(hap)]
    z = (rng.uniform_vector(12 * n_frag).reshape(n_frag, 12).sum(axis=1) - 6.0)
    ins = np.clip((400 + 50 * z).astype(np.int64), read_len, len(hap))
    st = np.floor(u[:n_frag] * (len(hap) - ins + 1)).astype(np.int64)
    cols = np.arange(read_len)
    m1 = hap[st[:, None] + cols[None, :]]
    s2 = st + ins - read_len
    m2 = _COMP[hap[s2[:, None] + cols[None, :]]][:, ::-1]
    ov1 = (st < keep_hi) & (st + read_len > keep_lo)
    ov2 = (s2 < keep_hi) & (s2 + read_len > keep_lo)
    keep = ov1 | ov2
    swap = u[n_frag:] < 0.5  # which mate is "first"/forward in the BAM is arbitrary: flip half of the pairs
    reads = np.empty((2 * n_frag, read_len), np.uint8)
    reads[0::2] = np.where(swap[:, None], m2, m1)
    reads[1::2] = np.where(swap[:, None], m1, m2)
    rev = np.zeros(2 * n_frag, np.uint8)
    rev[0::2] = swap
    rev[1::2] = ~swap
    # the aligner receives reads as stored in the BAM: reverse-strand reads are stored reverse-complemented,
    # i.e. in reference orientation
    flip = rev.astype(bool)
    reads[flip] = _COMP[reads[flip]][:, ::-1]
    frag = np.repeat(np.arange(n_frag, dtype=np.uint32) + frag0, 2)
    keep2 = np.repeat(keep, 2)
    reads, frag, rev = reads[keep2], frag[keep2], rev[keep2]
    code = np.zeros(256, dtype=np.int64)
    for k, c in enumerate(b"ACGT"):
        code[c] = k
    su = rng.uniform_vector(reads.size).reshape(reads.shape)
    sb = (rng.vector(reads.size) % np.uint64(3)).astype(np.int64).reshape(reads.shape)
    reads = np.where(su < sub_rate, _ACGT[(code[reads] + 1 + sb) & 3], reads)
    return reads, frag, rev


def mixed_sites(n_sites, seed=3, contig_len=None, read_len=150, depth=30.0, flank=150, sub_rate=0.01, margin=450,
                site_streams=False, indices=None, contig=None):
    """n_sites DEL/INS sites on one synthetic contig with 30x paired reads from a diploid genotype
    (0/0 : 0/1 : 1/1 = 1 : 2 : 1).  DEL length log-uniform 50..10000 (> 2*flank -> long-deletion template),
    INS length log-uniform 50..1000.  Returns list[SiteReads].

    site_streams=False: one random stream runs through all sites (the data sets of round 1).  site_streams=True: every
    site draws from its own stream seeded by (seed, site index), so any subset (`indices`) can be generated on its own --
    in parallel (mixed_sites_parallel) or per rank -- and is identical to the same sites of the full set."""
    rng = SplitMix64(seed)
    spacing = 14000
    contig_len = contig_len or (n_sites * spacing + 20000)
    if contig is None:
        contig = np.frombuffer(random_contig(seed * 7919 + 1, contig_len), dtype=np.uint8)
    cb = contig.tobytes() if not isinstance(contig, bytes) else contig
    contig = np.frombuffer(cb, dtype=np.uint8)
    out = []
    if indices is not None and not site_streams:
        raise ValueError("indices needs site_streams=True")
    for s in (range(n_sites) if indices is None else indices):
        if site_streams:
            rng = SplitMix64((seed * 0x9E3779B1 + 0x632BE59B * (s + 1)) & _MASK)
        center = 10000 + s * spacing
        is_del = rng.uniform() < 0.5
        gt = [0, 1, 1, 2][rng.below(4)]  # number of ALT alleles
        if is_del:
            dl = _log_uniform(rng, 50, 10000)
            start, end = center, center + dl - 1
            site = longdel_site(cb, start, end, flank) if dl > 2 * flank else del_site(cb, start, end, flank)
            lo, hi = start - flank - 1 - margin, end + flank + 1 + margin
            ref_hap = contig[lo:hi]
            alt_hap = np.concatenate([contig[lo:start - 1], contig[end:hi]])
            # target regions on each haplotype (graph span)
            if dl > 2 * flank:
                ref_keep = [(margin, margin + 2 * flank + 2), (len(ref_hap) - margin - 2 * flank - 2, len(ref_hap) - margin)]
            else:
                ref_keep = [(margin, len(ref_hap) - margin)]
            alt_keep = [(margin, len(alt_hap) - margin)]
        else:
            il = _log_uniform(rng, 50, 1000)
            ins = _ACGT[(rng.vector(il) & np.uint64(3)).astype(np.int64)]
            start = center
            site = ins_site(cb, start, ins.tobytes(), flank)
            lo, hi = start - flank - 1 - margin, start + flank + 1 + margin
            ref_hap = contig[lo:hi]
            alt_hap = np.concatenate([contig[lo:start], ins, contig[start:hi]])
            ref_keep = [(margin, len(ref_hap) - margin)]
            alt_keep = [(margin, len(alt_hap) - margin)]
        parts = []
        frag0 = 0
        for hap, keeps, copies in ((ref_hap, ref_keep, 2 - gt), (alt_hap, alt_keep, gt)):
            if copies == 0:
                continue
            for (klo, khi) in keeps:
                r, f, v = _paired_reads(rng, hap, klo, khi, depth * copies / 2.0, read_len, sub_rate, frag0)
                frag0 += len(r) // 2 + 1 + int(depth * len(hap) / (2.0 * read_len))
                parts.append((r, f, v))
        reads = np.concatenate([p[0] for p in parts]) if parts else np.zeros((0, read_len), np.uint8)
        frag = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, np.uint32)
        rev = np.concatenate([p[2] for p in parts]) if parts else np.zeros(0, np.uint8)
        out.append(SiteReads(site, reads, frag, rev, gt))
    return out


_POOL_ARGS = None


def _pool_sites(idx):
    n_sites, seed, kw, contig = _POOL_ARGS
    return mixed_sites(n_sites, seed, site_streams=True, indices=idx, contig=contig, **kw)


def mixed_sites_parallel(n_sites, seed=3, procs=8, **kw):
    """mixed_sites(..., site_streams=True) generated by `procs` forked workers (call BEFORE the process touches HIP: the
    workers are plain forks).  Same result as the serial call."""
    global _POOL_ARGS
    import multiprocessing as mp
    contig_len = n_sites * 14000 + 20000
    contig = random_contig(seed * 7919 + 1, contig_len)
    procs = max(1, min(int(procs), n_sites))
    if procs == 1:
        return mixed_sites(n_sites, seed, site_streams=True, contig=contig, **kw)
    chunks = [list(range(lo, min(n_sites, lo + 64))) for lo in range(0, n_sites, 64)]
    _POOL_ARGS = (n_sites, seed, kw, contig)
    try:
        with mp.get_context("fork").Pool(procs) as pool:
            parts = pool.map(_pool_sites, chunks)
    finally:
        _POOL_ARGS = None
    return [s for part in parts for s in part]


def long_node_site(seed, alt_len, flank=300, ref_mid=60):
    """BASELINE configs[4]-style graph: LF -> {REF (short), ALT (inline, 2-8 kb)} -> RF."""
    rng = SplitMix64(seed)
    def rnd(n):
        return _ACGT[(rng.vector(n) & np.uint64(3)).astype(np.int64)].tobytes()
    return Site("longalt", ["LF", "REF", "ALT", "RF"], [rnd(flank), rnd(ref_mid), rnd(alt_len), rnd(flank)],
                [(0, 1), (0, 2), (1, 3), (2, 3)],
                {(0, 1): ["REF"], (1, 3): ["REF"], (0, 2): ["ALT"], (2, 3): ["ALT"]},
                {"REF": [0, 1, 3], "ALT": [0, 2, 3]})
