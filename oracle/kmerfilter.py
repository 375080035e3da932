"""KmerFilter checkers -- TEST INFRASTRUCTURE ONLY.

* ``ref_kmer_filter``  : oracle/_ref/libpg_refcounts.so (the reference's graph-tools decodeGraphAlignment / numClipped /
                         extendPath + restated KmerIndex counts and filter control flow, oracle/ref_counts.cpp)
* ``port_kmer_filter`` : pure-Python restatement of
      KmerFilter::filterRead                src/c++/lib/paragraph/readfilters/KmerFilter.cpp:78-139
      KmerIndex + updateKmerCounts          GT!/src/graphalign/KmerIndex.cpp:76-141
      findMinCoveringKmerLength             GT!/src/graphalign/KmerIndexOperations.cpp:77-113
  pinned on src/c++/test/test_readfilter.cpp:90-166 (tests/test_kmerfilter_oracle.py).

reads: list of (graph_pos, graph_cigar, bases) with the bases as the aligner left them (reverse-complemented when
the alignment is on the reverse strand).  Returns (k used, [(filtered, message)]).
"""
import ctypes as C

import numpy as np

from . import counts as oc
from .pathalign import _kmer_paths, _path_seq


def _index(nodes, edges, k):
    n = len(nodes)
    succ = [sorted({t for f, t in edges if f == i}) for i in range(n)]
    index = {}
    for p in _kmer_paths(nodes, succ, k):
        index.setdefault(_path_seq(nodes, p), []).append(p)
    node_counts, edge_counts = {}, {}
    for paths in index.values():
        if len(paths) != 1:
            continue
        nl = paths[0][1]
        for i, nd in enumerate(nl):
            node_counts[nd] = node_counts.get(nd, 0) + 1
            if i:
                edge_counts[(nl[i - 1], nd)] = edge_counts.get((nl[i - 1], nd), 0) + 1
    return index, node_counts, edge_counts, succ


def min_covering_kmer_length(nodes, edges, need):
    for k in range(10, 64):
        _, nc, ec, succ = _index(nodes, edges, k)
        ok = True
        for nd in range(len(nodes)):
            if nc.get(nd, 0) < need:
                ok = False
                break
            if any(ec.get((nd, s), 0) < need for s in succ[nd]):
                ok = False
                break
        if ok:
            return k
    return -1


def port_kmer_filter(nodes, edges, kmer_len, reads):
    if kmer_len < 0:
        kmer_len = min_covering_kmer_length(nodes, edges, -kmer_len)
        if kmer_len < 0:
            return -1, []
    index, node_counts, _, _ = _index(nodes, edges, kmer_len)
    g = oc.CountGraph(nodes, edges)
    out = []
    for pos, cigar, bases in reads:
        try:
            aln = oc.decode(g, pos, cigar)
        except oc.DecodeError:
            out.append((True, "kmer_nomapping"))
            continue
        sc_left = aln[0][1]["S"]
        sc_right = aln[-1][1]["S"]
        if len(bases) - sc_left - sc_right < kmer_len:
            out.append((True, "kmer_tooshort"))
            continue
        kmers = {bases[p:p + kmer_len] for p in range(sc_left, len(bases) - sc_right - kmer_len + 1)}
        supported = [nd for nd, _ in aln if node_counts.get(nd, 0) > 0]
        not_covered = set(supported)
        passed = False
        for km in kmers:
            paths = index.get(km)
            if paths is not None and len(paths) == 1:
                for nd in paths[0][1]:
                    not_covered.discard(nd)
                passed = passed or not not_covered
        if passed:
            out.append((False, ""))
        else:
            out.append((True, "kmer_uncov" + "".join("_%d" % nd for nd in supported if nd in not_covered)))
    return kmer_len, out


def ref_kmer_filter(nodes, edges, kmer_len, reads):
    rc = oc.RefCounts()
    L = rc.L
    u32p = C.POINTER(C.c_uint32)
    L.pgrefc_kmer_filter.restype = C.c_int
    L.pgrefc_kmer_filter.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_int32), u32p, C.c_char_p, u32p, C.c_char_p,
                                     C.c_char_p, C.c_char_p, C.c_int]
    seq_off = np.zeros(len(nodes) + 1, dtype=np.uint32)
    seq_off[1:] = np.cumsum([len(s) for s in nodes])
    frm = np.array([e[0] for e in edges] or [0], dtype=np.uint32)
    to = np.array([e[1] for e in edges] or [0], dtype=np.uint32)
    loff = np.zeros(len(edges) + 1, dtype=np.uint32)
    lids = np.zeros(1, dtype=np.uint32)
    names = (C.c_char_p * 1)(b"x")

    def P(a):
        return a.ctypes.data_as(u32p)
    g = L.pgrefc_graph_create(len(nodes), P(seq_off), "".join(nodes).encode(), len(edges), P(frm), P(to), P(loff), P(lids), 0, names)
    if not g:
        raise RuntimeError("pgrefc_graph_create failed")
    n = len(reads)
    pos = np.array([r[0] for r in reads] or [0], dtype=np.int32)
    coff = np.zeros(n + 1, dtype=np.uint32)
    coff[1:] = np.cumsum([len(r[1]) for r in reads])
    boff = np.zeros(n + 1, dtype=np.uint32)
    boff[1:] = np.cumsum([len(r[2]) for r in reads])
    out = C.create_string_buffer(max(n, 1))
    stride = 256
    msgs = C.create_string_buffer(max(n, 1) * stride)
    k = L.pgrefc_kmer_filter(g, kmer_len, n, pos.ctypes.data_as(C.POINTER(C.c_int32)), P(coff), "".join(r[1] for r in reads).encode(),
                             P(boff), "".join(r[2] for r in reads).encode(), out, msgs, stride)
    L.pgrefc_graph_destroy(g)
    if k < 0:
        return -1, []
    return k, [(out.raw[i] != 0, msgs.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode()) for i in range(n)]
