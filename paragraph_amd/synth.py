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
