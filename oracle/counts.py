"""Count-path checkers -- TEST INFRASTRUCTURE ONLY.

* ``RefCounts``  : oracle/_ref/libpg_refcounts.so = the reference's own graph-tools library (Graph, Path
                   validity, PathFamily::containsPath, decodeGraphAlignment, Alignment counters) + a restated
                   glue (oracle/ref_counts.cpp).
* ``port_count_site`` : pure-Python restatement of the same path (small cases), following
      decodeGraphAlignment / Alignment / Operation   GT!/src/graphalign/GraphAlignmentOperations.cpp:67-127,
                                                     LinearAlignment.cpp:49-131, Operation.cpp:55-111
      Path validity (what makes decode throw)        GT!/src/graphcore/Path.cpp:86-190
      PathFamily::containsPath                       GT!/src/graphcore/PathFamily.cpp:89-108
      NonUniq / BadAlign / filter chain              src/c++/lib/paragraph/readfilters/*.hh, ReadFilter.cpp:43-90
      nodefilter / edgefilter                        src/c++/lib/paragraph/Disambiguation.cpp:212-296
      disambiguateReads                              src/c++/lib/paragraph/Disambiguation.cpp:82-142
      fragments + counts                             src/c++/lib/common/Fragment.cpp:34-69,141-181,
                                                     src/c++/lib/paragraph/ReadCounting.cpp:52-127
"""
import ctypes as C
import math
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_PATH = os.path.join(_HERE, "_ref", "libpg_refcounts.so")


def have_ref():
    return os.path.exists(REF_PATH)


class Params(C.Structure):
    _fields_ = [("remove_nonuniq", C.c_int32), ("bad_align_frac", C.c_double), ("use_support_filters", C.c_int32)]


class CountGraph:
    """nodes: list[str]; edges: list[(from,to)]; edge_labels: {(from,to): [label,...]}; labels: ordered list."""

    def __init__(self, nodes, edges, edge_labels=None, labels=None):
        self.nodes = list(nodes)
        self.edges = [tuple(e) for e in edges]
        self.edge_labels = {tuple(k): list(v) for k, v in (edge_labels or {}).items()}
        if labels is None:
            labels = sorted({l for v in self.edge_labels.values() for l in v})
        self.labels = list(labels)


def _round_half_away(x):
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


_NODE_RE = re.compile(r"(\d+)\[([^\]]*)\]")
_OP_RE = re.compile(r"(\d+)([A-Za-z])")
_REF_OPS = "MXND"
_QRY_OPS = "MXNIS"


class DecodeError(Exception):
    pass


def decode(graph, pos, cigar):
    """-> list of (node_id, counts dict) or raises DecodeError where graph-tools would throw."""
    if not cigar:
        raise DecodeError("empty")
    consumed = "".join(m.group(0) for m in _NODE_RE.finditer(cigar))
    if consumed != cigar:
        raise DecodeError("malformed")
    out = []
    for m in _NODE_RE.finditer(cigar):
        node = int(m.group(1))
        body = m.group(2)
        if "".join(x.group(0) for x in _OP_RE.finditer(body)) != body:
            raise DecodeError("malformed node cigar")
        c = {"M": 0, "X": 0, "N": 0, "I": 0, "D": 0, "S": 0}
        for x in _OP_RE.finditer(body):
            if x.group(2) not in c:
                raise DecodeError("unknown op")
            c[x.group(2)] += int(x.group(1))
        c["rlen"] = sum(c[o] for o in _REF_OPS)
        c["qlen"] = sum(c[o] for o in _QRY_OPS)
        out.append((node, c))
    nodes = [n for n, _ in out]
    if any(n >= len(graph.nodes) for n in nodes):
        raise DecodeError("node out of range")
    # Path validity (Path.cpp:86-190)
    last_start = pos if len(out) == 1 else 0
    end_pos = last_start + out[-1][1]["rlen"] - 1
    if pos < 0 or pos >= len(graph.nodes[nodes[0]]):
        raise DecodeError("first position")
    if end_pos < 0 or end_pos >= len(graph.nodes[nodes[-1]]):
        raise DecodeError("last position")
    if any(a > b for a, b in zip(nodes, nodes[1:])):
        raise DecodeError("unordered")
    if len(nodes) == 1 and pos > end_pos:
        raise DecodeError("positions")
    eset = set(graph.edges)
    for a, b in zip(nodes, nodes[1:]):
        if (a, b) not in eset:  # Path::isPathConnected = hasEdge for every consecutive pair
            raise DecodeError("disconnected")
    return out


def _node_filter(graph, aln, read_len, node_id):
    for n, c in aln:
        if n == node_id:
            short = len(graph.nodes[node_id]) < read_len // 2
            nonmatch = c["X"] + c["S"]
            indel = c["I"] + c["D"]
            if short and (nonmatch > 0 or indel > 0):
                return False
            return nonmatch + indel <= read_len // 2
    return False


def _edge_filter(graph, aln, read_len, n1, n2):
    prev = None
    for n, c in aln:
        if prev is not None and prev[0] == n1 and n == n2:
            p = prev[1]
            mno = read_len // 10 + 1
            ok = p["M"] >= min(p["rlen"], mno) and c["M"] >= min(c["rlen"], mno)
            if ok:
                ok = p["qlen"] < p["rlen"] * 2 and c["qlen"] < c["rlen"] * 2
            if ok:
                ok = p["M"] >= min(len(graph.nodes[n1]), mno) and c["M"] >= min(len(graph.nodes[n2]), mno)
            return ok
        prev = (n, c)
    return False


def _contains_path(fam_edges, path):
    out_nodes = {a for a, _ in fam_edges}
    in_nodes = {b for _, b in fam_edges}
    matched = 0
    for a, b in zip(path, path[1:]):
        if (a, b) in fam_edges:
            matched += 1
        elif a in out_nodes or b in in_nodes:
            return False
    return matched > 0


def port_count_site(graph, reads, remove_nonuniq=True, bad_align_frac=0.8, use_support_filters=True):
    """reads: list of dicts {pos, cigar, aligned, unique, graph_reverse, read_len, fragment}.
    Returns dict(status, nodes, edges, labels (per read, as sets), node_counts, edge_counts, seq_counts, rc)."""
    rc = 0
    n = len(reads)
    status = [0] * n
    alns = [None] * n
    for i, r in enumerate(reads):
        if not r["aligned"]:
            continue
        status[i] = 1
        bad = bool(remove_nonuniq and not r["unique"])
        if not bad:
            try:
                aln = decode(graph, r["pos"], r["cigar"])
                alns[i] = aln
                qlen = sum(c["qlen"] for _, c in aln)
                clipped = sum(c["S"] for _, c in aln)
                bad = (qlen - clipped) < _round_half_away(bad_align_frac * qlen)
            except DecodeError:
                rc = -1
                bad = True
        if bad:
            status[i] = 2
    rn, re_, rl = [set() for _ in range(n)], [set() for _ in range(n)], [set() for _ in range(n)]
    lab_edges = {l: {e for e, ls in graph.edge_labels.items() if l in ls} for l in graph.labels}
    for i, r in enumerate(reads):
        if status[i] != 1:
            continue
        aln = alns[i]
        if aln is None:
            rc = -1
            continue
        path = [nd for nd, _ in aln]
        overl = set()
        prev = None
        for nd in path:
            if prev is not None and (not use_support_filters or _edge_filter(graph, aln, r["read_len"], prev, nd)):
                re_[i].add((prev, nd))
                overl.update(graph.edge_labels.get((prev, nd), []))
            prev = nd
            if not use_support_filters or _node_filter(graph, aln, r["read_len"], nd):
                rn[i].add(nd)
        for l in overl:
            if _contains_path(lab_edges[l], path):
                rl[i].add(l)
    frags = {}
    order = []
    for i, r in enumerate(reads):
        if status[i] != 1:
            continue
        f = frags.get(r["fragment"])
        if f is None:
            f = frags[r["fragment"]] = {"n": 0, "fwd": 0, "rev": 0, "nodes": set(), "edges": set(), "labels": set()}
            order.append(f)
        f["n"] += 1
        f["rev" if r["graph_reverse"] else "fwd"] += 1
        f["nodes"] |= rn[i]
        f["edges"] |= re_[i]
        f["labels"] |= rl[i]
    node_counts = np.zeros((len(graph.nodes), 4), dtype=np.uint64)
    edge_counts = np.zeros((len(graph.edges), 4), dtype=np.uint64)
    eidx = {e: k for k, e in enumerate(graph.edges)}
    seq_counts = {}
    for f in order:
        inc = np.array([1, f["n"], f["fwd"], f["rev"]], dtype=np.uint64)
        for nd in f["nodes"]:
            node_counts[nd] += inc
        for e in f["edges"]:
            edge_counts[eidx[e]] += inc
        if f["labels"]:
            key = ",".join(sorted(f["labels"]))
            seq_counts[key] = seq_counts.get(key, np.zeros(4, dtype=np.uint64)) + inc
    return {"status": status, "nodes": rn, "edges": re_, "labels": rl, "node_counts": node_counts,
            "edge_counts": edge_counts, "seq_counts": {k: [int(x) for x in v] for k, v in seq_counts.items()},
            "rc": rc}


class RefCounts:
    def __init__(self):
        if not have_ref():
            raise FileNotFoundError(REF_PATH)
        L = C.CDLL(REF_PATH)
        u32p, i32p, u8p, u64p = (C.POINTER(t) for t in (C.c_uint32, C.c_int32, C.c_uint8, C.c_uint64))
        L.pgrefc_graph_create.restype = C.c_void_p
        L.pgrefc_graph_create.argtypes = [C.c_uint32, u32p, C.c_char_p, C.c_uint32, u32p, u32p, u32p, u32p, C.c_uint32,
                                          C.POINTER(C.c_char_p)]
        L.pgrefc_graph_destroy.restype = None
        L.pgrefc_graph_destroy.argtypes = [C.c_void_p]
        L.pgrefc_count_site.restype = C.c_int
        L.pgrefc_count_site.argtypes = [C.c_void_p, C.c_uint32, i32p, u32p, C.c_char_p, u8p, u8p, u8p, u32p, u32p,
                                        C.POINTER(Params), u8p, u64p, u64p, u64p, u64p, u64p, u32p, u64p, u64p,
                                        C.c_uint32]
        L.pgrefc_pair_length.restype = C.c_int
        L.pgrefc_pair_length.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, u64p, u64p]
        self.L = L

    def _graph(self, graph):
        def P(a, t):
            return a.ctypes.data_as(C.POINTER(t))
        n_nodes = len(graph.nodes)
        seq_off = np.zeros(n_nodes + 1, dtype=np.uint32)
        seq_off[1:] = np.cumsum([len(s) for s in graph.nodes])
        frm = np.array([e[0] for e in graph.edges] or [0], dtype=np.uint32)
        to = np.array([e[1] for e in graph.edges] or [0], dtype=np.uint32)
        loff = np.zeros(len(graph.edges) + 1, dtype=np.uint32)
        lids = np.array([0], dtype=np.uint32)
        names = (C.c_char_p * 1)(b"")
        g = self.L.pgrefc_graph_create(n_nodes, P(seq_off, C.c_uint32), "".join(graph.nodes).encode(), len(graph.edges),
                                       P(frm, C.c_uint32), P(to, C.c_uint32), P(loff, C.c_uint32), P(lids, C.c_uint32), 0, names)
        if not g:
            raise RuntimeError("pgrefc_graph_create failed")
        return g

    def pair_lengths(self, graph, pairs):
        """Graph length of two-read fragments as common::Fragment::addRead computes it on the reference's own
        graphtools::GraphCoordinates (Fragment.cpp:70-95): pairs = [(pos1, cigar1, pos2, cigar2)] -> [length | None]."""
        g = self._graph(graph)
        out = []
        try:
            span = (C.c_uint64 * 4)()
            length = C.c_uint64(0)
            for pos1, cigar1, pos2, cigar2 in pairs:
                if self.L.pgrefc_pair_length(g, pos1, cigar1.encode(), pos2, cigar2.encode(), span, C.byref(length)) != 0:
                    raise RuntimeError("pgrefc_pair_length failed on %r" % ((pos1, cigar1, pos2, cigar2),))
                out.append(None if length.value == 0xFFFFFFFFFFFFFFFF else int(length.value))
        finally:
            self.L.pgrefc_graph_destroy(g)
        return out

    def count_site(self, graph, reads, remove_nonuniq=True, bad_align_frac=0.8, use_support_filters=True):
        def P(a, t):
            return a.ctypes.data_as(C.POINTER(t))
        n_nodes = len(graph.nodes)
        seq_off = np.zeros(n_nodes + 1, dtype=np.uint32)
        seq_off[1:] = np.cumsum([len(s) for s in graph.nodes])
        frm = np.array([e[0] for e in graph.edges] or [0], dtype=np.uint32)
        to = np.array([e[1] for e in graph.edges] or [0], dtype=np.uint32)
        lab_idx = {l: k for k, l in enumerate(graph.labels)}
        loff, lids = [0], []
        for e in graph.edges:
            lids.extend(lab_idx[l] for l in graph.edge_labels.get(e, []))
            loff.append(len(lids))
        loff = np.array(loff, dtype=np.uint32)
        lids = np.array(lids or [0], dtype=np.uint32)
        names = (C.c_char_p * max(1, len(graph.labels)))(*[l.encode() for l in graph.labels])
        g = self.L.pgrefc_graph_create(n_nodes, P(seq_off, C.c_uint32), "".join(graph.nodes).encode(),
                                       len(graph.edges), P(frm, C.c_uint32), P(to, C.c_uint32), P(loff, C.c_uint32),
                                       P(lids, C.c_uint32), len(graph.labels), names)
        if not g:
            raise RuntimeError("pgrefc_graph_create failed")
        n = len(reads)
        pos = np.array([r["pos"] for r in reads] or [0], dtype=np.int32)
        coff = np.zeros(n + 1, dtype=np.uint32)
        coff[1:] = np.cumsum([len(r["cigar"]) for r in reads])
        cig = "".join(r["cigar"] for r in reads).encode()
        al = np.array([1 if r["aligned"] else 0 for r in reads] or [0], dtype=np.uint8)
        un = np.array([1 if r["unique"] else 0 for r in reads] or [0], dtype=np.uint8)
        rv = np.array([1 if r["graph_reverse"] else 0 for r in reads] or [0], dtype=np.uint8)
        rl = np.array([r["read_len"] for r in reads] or [0], dtype=np.uint32)
        fr = np.array([r["fragment"] for r in reads] or [0], dtype=np.uint32)
        prm = Params(1 if remove_nonuniq else 0, bad_align_frac, 1 if use_support_filters else 0)
        st = np.zeros(max(n, 1), dtype=np.uint8)
        om, oe = (np.zeros(max(n, 1), dtype=np.uint64) for _ in range(2))
        ol = np.zeros((max(n, 1), 4), dtype=np.uint64)  # label sets: four words per read (up to 256 labels)
        nc = np.zeros((n_nodes, 4), dtype=np.uint64)
        ec = np.zeros((max(1, len(graph.edges)), 4), dtype=np.uint64)
        nseq = C.c_uint32()
        smask = np.zeros((256, 4), dtype=np.uint64)
        scnt = np.zeros((256, 4), dtype=np.uint64)
        rc = self.L.pgrefc_count_site(g, n, P(pos, C.c_int32), P(coff, C.c_uint32), cig, P(al, C.c_uint8),
                                      P(un, C.c_uint8), P(rv, C.c_uint8), P(rl, C.c_uint32), P(fr, C.c_uint32),
                                      C.byref(prm), P(st, C.c_uint8), P(om, C.c_uint64), P(oe, C.c_uint64),
                                      P(ol, C.c_uint64), P(nc, C.c_uint64), P(ec, C.c_uint64), C.byref(nseq),
                                      P(smask, C.c_uint64), P(scnt, C.c_uint64), 256)
        self.L.pgrefc_graph_destroy(g)

        def bits(m, universe):
            if isinstance(m, np.ndarray):  # a label set of four words
                m = sum(int(w) << (64 * k) for k, w in enumerate(m))
            return {universe[k] for k in range(len(universe)) if (int(m) >> k) & 1}
        seqs = {}
        for k in range(nseq.value):
            key = ",".join(sorted(bits(smask[k], graph.labels)))
            seqs[key] = [int(x) for x in scnt[k]]
        return {"status": [int(x) for x in st[:n]],
                "nodes": [bits(om[i], list(range(n_nodes))) for i in range(n)],
                "edges": [bits(oe[i], graph.edges) for i in range(n)],
                "labels": [bits(ol[i], graph.labels) for i in range(n)],
                "node_counts": nc, "edge_counts": ec[:len(graph.edges)], "seq_counts": seqs, "rc": rc}
