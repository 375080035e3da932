"""PathAligner checkers -- TEST INFRASTRUCTURE ONLY.

* ``ref_path_align``  : oracle/_ref/libpg_refcounts.so (the reference's graph-tools extendPath /
                        extendPathMatching / projectAlignmentOntoGraph + restated control flow, ref_counts.cpp)
* ``port_path_align`` : pure-Python restatement of
      KmerIndex construction        GT!/src/graphalign/KmerIndex.cpp:76-116 (+ extendPathEnd, PathOperations.cpp:70-101)
      extendPathMatching            GT!/src/graphcore/PathOperations.cpp:117-271
      PathAligner::alignRead        src/c++/lib/grm/PathAligner.cpp:75-164
      projectAlignmentOntoGraph for an all-M alignment   GT!/src/graphalign/GraphAlignmentOperations.cpp:130-164
"""
import ctypes as C

import numpy as np

from .counts import REF_PATH, have_ref  # noqa: F401


class PathResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("status", "graph_pos", "score", "mapq", "unique", "is_graph_reverse",
                                         "anchored", "cigar_len")]


def _rc(s):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    return "".join(comp.get(c, "N") for c in reversed(s))


def _kmer_paths(nodes, succ, k):
    """All length-k paths: (start_pos, [node ids], end_pos), in KmerIndex's enumeration order."""
    out = []

    def extend_end(nl, start, endpos, need):
        last = nl[-1]
        room = len(nodes[last]) - endpos - 1
        if need <= room:
            return [(start, list(nl), endpos + need)]
        res = []
        for s in succ[last]:
            res.extend(extend_end(nl + [s], start, 0, need - room - 1))
        return res
    for n, seq in enumerate(nodes):
        for pos in range(len(seq)):
            out.extend(extend_end([n], pos, pos, k - 1))
    return out


def _path_seq(nodes, p):
    start, nl, end = p
    if len(nl) == 1:
        return nodes[nl[0]][start:end + 1]
    return nodes[nl[0]][start:] + "".join(nodes[x] for x in nl[1:-1]) + nodes[nl[-1]][:end + 1]


def _path_len(nodes, p):
    return len(_path_seq(nodes, p))


def _extend_matching(nodes, succ, pred, p, query, qpos):
    start, nl, end = p
    nl = list(nl)
    # ---- extendPathEndMatching
    pos_in_query = qpos + _path_len(nodes, p)
    node = nl[-1]
    pos_in_node = end + 1
    moved = True
    while moved:
        moved = False
        seq = nodes[node]
        while pos_in_query < len(query) and pos_in_node < len(seq) and query[pos_in_query] == seq[pos_in_node]:
            moved = True
            pos_in_node += 1
            pos_in_query += 1
        if pos_in_node >= len(seq):
            ss = succ[node]
            nlm, best, cur = 0, 0, 0
            msz = min([len(nodes[s]) for s in ss], default=1 << 60)
            for s in ss:
                q = 0
                sseq = nodes[s]
                while q < msz and pos_in_query + q < len(query) and sseq[q] == query[pos_in_query + q]:
                    q += 1
                if q > best:
                    best, cur, nlm = q, s, 1
                elif q == best:
                    nlm += 1
            if best == 0 or nlm != 1:
                break
            nl.append(cur)
            pos_in_query += best
            pos_in_node = best
            node = cur
            moved = True
    end = pos_in_node - 1
    # ---- extendPathStartMatching
    pos_in_query = qpos
    node = nl[0]
    pos_in_node = start
    moved = True
    while moved:
        moved = False
        seq = nodes[node]
        while pos_in_query > 0 and pos_in_node > 0 and query[pos_in_query - 1] == seq[pos_in_node - 1]:
            moved = True
            pos_in_node -= 1
            pos_in_query -= 1
        if pos_in_node == 0:
            ps = pred[node]
            nlm, best, cur = 0, 0, 0
            msz = min([len(nodes[s]) for s in ps], default=1 << 60)
            for s in ps:
                pseq = nodes[s]
                pp = len(pseq)
                ml = 0
                while pp > len(pseq) - msz and pos_in_query - ml > 0 and pseq[pp - 1] == query[pos_in_query - ml - 1]:
                    pp -= 1
                    ml += 1
                if ml > best:
                    best, cur, nlm = ml, s, 1
                elif ml == best:
                    nlm += 1
            if best == 0 or nlm != 1:
                break
            nl.insert(0, cur)
            pos_in_query -= best
            node = cur
            pos_in_node = len(nodes[node]) - best
            moved = True
    return (pos_in_node, nl, end), pos_in_query


def port_path_align(nodes, edges, reads, kmer_len=32):
    n = len(nodes)
    succ = [sorted({t for f, t in edges if f == i}) for i in range(n)]
    pred = [sorted({f for f, t in edges if t == i}) for i in range(n)]
    index = {}
    for p in _kmer_paths(nodes, succ, kmer_len):
        index.setdefault(_path_seq(nodes, p), []).append(p)
    out = []
    for read in reads:
        res = {"status": 0, "graph_pos": 0, "score": 0, "mapq": 0, "unique": False, "is_graph_reverse": False,
               "anchored": False, "cigar": ""}
        L = len(read)
        if L >= kmer_len:
            matches = []
            for strand in (0, 1):
                rb = _rc(read) if strand else read
                pos = 0
                while pos + kmer_len <= len(rb):
                    paths = index.get(rb[pos:pos + kmer_len])
                    if paths is not None and len(paths) == 1:
                        ext, qpos = _extend_matching(nodes, succ, pred, paths[0], rb, pos)
                        matches.append((qpos, ext, bool(strand)))
                        pos = qpos + _path_len(nodes, ext)
                    pos += 1
            res["anchored"] = bool(matches)
            full = [m for m in matches if _path_len(nodes, m[1]) == L]
            if full:
                qpos, (start, nl, end), rev = full[0]
                parts = []
                for i, nd in enumerate(nl):
                    a = start if i == 0 else 0
                    b = end if i == len(nl) - 1 else len(nodes[nd]) - 1
                    parts.append("%d[%dM]" % (nd, b - a + 1))
                res.update(status=1, graph_pos=start, score=L, is_graph_reverse=rev, unique=len(full) == 1,
                           mapq=60 if len(full) == 1 else 0, cigar="".join(parts))
        out.append(res)
    return out


def ref_path_align(nodes, edges, reads, kmer_len=32):
    from .counts import RefCounts
    rc = RefCounts()
    L = rc.L
    u32p = C.POINTER(C.c_uint32)
    L.pgrefc_path_align.restype = C.c_int
    L.pgrefc_path_align.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, u32p, C.c_char_p, C.POINTER(PathResult), C.c_char_p,
                                    C.c_int]
    n_nodes = len(nodes)
    seq_off = np.zeros(n_nodes + 1, dtype=np.uint32)
    seq_off[1:] = np.cumsum([len(s) for s in nodes])
    frm = np.array([e[0] for e in edges] or [0], dtype=np.uint32)
    to = np.array([e[1] for e in edges] or [0], dtype=np.uint32)
    loff = np.zeros(len(edges) + 1, dtype=np.uint32)
    lids = np.zeros(1, dtype=np.uint32)
    names = (C.c_char_p * 1)(b"x")

    def P(a):
        return a.ctypes.data_as(u32p)
    g = L.pgrefc_graph_create(n_nodes, P(seq_off), "".join(nodes).encode(), len(edges), P(frm), P(to), P(loff), P(lids), 0,
                              names)
    if not g:
        raise RuntimeError("pgrefc_graph_create failed")
    off = np.zeros(len(reads) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(r) for r in reads])
    res = (PathResult * max(1, len(reads)))()
    stride = 1024
    cig = C.create_string_buffer(max(1, len(reads)) * stride)
    L.pgrefc_path_align(g, kmer_len, len(reads), P(off), "".join(reads).encode(), res, cig, stride)
    L.pgrefc_graph_destroy(g)
    out = []
    for i in range(len(reads)):
        r = res[i]
        out.append({"status": r.status, "graph_pos": r.graph_pos, "score": r.score, "mapq": r.mapq, "unique": bool(r.unique),
                    "is_graph_reverse": bool(r.is_graph_reverse), "anchored": bool(r.anchored),
                    "cigar": cig.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode()})
    return out
