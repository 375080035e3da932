"""KmerAligner checker -- TEST INFRASTRUCTURE ONLY (pure Python restatement; small cases).

Follows
    grm::KmerAligner<K>                 src/c++/lib/grm/KmerAligner.cpp:120-133 (makeKmers), 135-177 (BasicPath),
                                        246-303 (seed merge-join + bounded heap), 305-319 (setGraph),
                                        321-472 (CIGAR construction), 478-538 (pickBest, alignRead)
    oligo::KmerGenerator / Translator   src/c++/include/oligo/KmerGenerator.hh:56-154, Nucleotides.hh:59-...
    std::push_heap / std::pop_heap      libstdc++ bits/stl_heap.h (the candidate heap's element ORDER decides which of
                                        several equally good candidates is reported, so it is restated literally)

The reference translation unit cannot be compiled in this image (oligo/Kmer.hh needs Boost.MPL); this restatement
is pinned on the reference's own unit test (src/c++/test/test_kmeraligner.cpp:149-193, KmerAligner<10>).
"""

_VAL = {c: i for i, c in enumerate("ACGT")}
_VAL.update({c.lower(): v for c, v in list(_VAL.items())})


def _rc(s):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    return "".join(comp.get(c, "N") for c in reversed(s))


def make_kmers(seq, k):
    """All windows of k consecutive valid bases, as sorted (kmer, position) pairs (KmerGenerator + std::sort)."""
    out = []
    run = 0
    val = 0
    mask = (1 << (2 * k)) - 1
    for i, c in enumerate(seq):
        v = _VAL.get(c)
        if v is None:
            run = 0
            val = 0
            continue
        val = ((val << 2) | v) & mask
        run += 1
        if run >= k:
            out.append((val, i - k + 1))
    out.sort()
    return out


def _push_heap(a, less):
    value = a[-1]
    hole = len(a) - 1
    parent = (hole - 1) // 2
    while hole > 0 and less(a[parent], value):
        a[hole] = a[parent]
        hole = parent
        parent = (hole - 1) // 2
    a[hole] = value


def _pop_heap(a, less):
    """std::pop_heap: moves the top to the back; the caller then pop_back()s."""
    value = a[-1]
    a[-1] = a[0]
    n = len(a) - 1
    hole = 0
    second = 0
    while second < (n - 1) // 2:
        second = 2 * (second + 1)
        if less(a[second], a[second - 1]):
            second -= 1
        a[hole] = a[second]
        hole = second
    if (n & 1) == 0 and second == (n - 2) // 2:
        second = 2 * (second + 1)
        a[hole] = a[second - 1]
        hole = second - 1
    # __push_heap(first, hole, 0, value)
    parent = (hole - 1) // 2
    while hole > 0 and less(a[parent], value):
        a[hole] = a[parent]
        hole = parent
        parent = (hole - 1) // 2
    a[hole] = value


def _cigar_op(ref, rd):
    return "M" if ref == rd else ("N" if ref == "N" or rd == "N" else "X")


class _Path:
    def __init__(self, pid, nodes, node_ids, k):
        self.pid = pid
        self.node_ids = list(node_ids)
        self.starts = []
        s = 0
        for n in node_ids:
            self.starts.append((s, n))
            s += len(nodes[n])
        self.seq = "".join(nodes[n] for n in node_ids)
        self.kmers = make_kmers(self.seq, k)

    def find_start(self, pos):
        idx = 0
        for i, (s, _) in enumerate(self.starts):
            if s <= pos:
                idx = i
        return idx


def _update_alignment(nodes, path, pos, reverse, bases, rv_bases, bam_reverse):
    seq = rv_bases if reverse else bases
    L = len(bases)
    ref = path.seq
    left = 0
    while left < L and ref[pos + left] == "N":
        left += 1
    pos2 = pos + left
    right = 0
    while right < L - left and ref[pos + L - 1 - right] == "N":
        right += 1
    si = path.find_start(pos2)
    start = pos2 - path.starts[si][0]
    length_left = L - left - right
    this_start = start
    cigar = []
    score = 0
    it = left
    lc, rcl = left, right
    i = si
    while i < len(path.starts) and length_left > 0:
        this_length = length_left
        if i + 1 < len(path.starts):
            this_length = min(length_left, path.starts[i + 1][0] - path.starts[i][0] - this_start)
        if this_length > 0:
            r0 = this_start + path.starts[i][0]
            bit = []
            last, ln = None, 0
            for j in range(this_length):
                op = _cigar_op(ref[r0 + j], seq[it + j])
                if op != last:
                    if ln:
                        bit.append("%d%s" % (ln, last))
                        if last == "M":
                            score += ln
                    last, ln = op, 0
                ln += 1
            if ln:
                bit.append("%d%s" % (ln, last))
                if last == "M":
                    score += ln
            it += this_length
            s = "%d[" % path.starts[i][1]
            if lc:
                s += "%dS" % lc
                lc = 0
            s += "".join(bit)
            if rcl and this_length == length_left:
                s += "%dS" % rcl
            s += "]"
            cigar.append(s)
        length_left -= this_length
        i += 1
        this_start = 0
    return {"graph_pos": start, "cigar": "".join(cigar), "score": score,
            "is_graph_reverse": (not bam_reverse) if reverse else bam_reverse, "used_reverse": reverse}


def port_kmer_align(nodes, paths, reads, k=16, bam_reverse=None):
    """nodes: list[str]; paths: list of node-id lists; returns per read dict(status 0 UNMAPPED / 1 MAPPED / 2 BAD_ALIGN, ...)."""
    P = [_Path(i, nodes, p, k) for i, p in enumerate(paths)]
    cap = len(P) + 2

    def less(x, y):
        return x[3] < y[3]
    out = []
    for ri, bases in enumerate(reads):
        br = bool(bam_reverse[ri]) if bam_reverse is not None else False
        res = {"status": 0, "graph_pos": 0, "cigar": "", "score": 0, "mapq": 0, "unique": False, "is_graph_reverse": False,
               "used_reverse": False}
        rv = _rc(bases)
        fw_k = make_kmers(bases, k) if bases else []
        rv_k = make_kmers(rv, k) if bases else []
        cands = []
        for path in P:
            for reverse, seq, sk in ((False, bases, fw_k), (True, rv, rv_k)):
                offs = set()
                pk = path.kmers
                pi = 0
                for km, sp in sk:
                    while pi < len(pk) and pk[pi][0] < km:
                        pi += 1
                    while pi < len(pk) and pk[pi][0] == km:
                        off = pk[pi][1] - sp
                        if 0 <= off and len(path.seq) >= off + len(seq):
                            offs.add(off)
                        pi += 1
                for off in sorted(offs):
                    mm = sum(1 for a, b in zip(seq, path.seq[off:off + len(seq)]) if a != b)
                    cands.append((path.pid, off, reverse, mm))
                    _push_heap(cands, less)
                    if len(cands) == cap:
                        _pop_heap(cands, less)
                        cands.pop()
        if cands:
            bi = min(range(len(cands)), key=lambda i: (cands[i][3], i))
            best = cands[bi]
            if best[3] <= 2:
                aln = _update_alignment(nodes, P[best[0]], best[1], best[2], bases, rv, br)
                res.update(aln)
                res.update(status=1, mapq=60, unique=True)
                i = bi + 1
                while i < len(cands):
                    si = min(range(i, len(cands)), key=lambda j: (cands[j][3], j))
                    sb = cands[si]
                    if sb[3] != best[3]:
                        break
                    a2 = _update_alignment(nodes, P[sb[0]], sb[1], sb[2], bases, rv, br)
                    if a2["cigar"] != res["cigar"] or a2["graph_pos"] != res["graph_pos"]:
                        res.update(status=2, mapq=0, unique=False)
                        break
                    i = si + 1
        out.append(res)
    return out
