"""ctypes loader of oracle/_ref/libpg_refstats.so -- TEST INFRASTRUCTURE ONLY: the reference's own
paragraph::summarizeAlignments (src/c++/lib/paragraph/GraphSummaryStatistics.cpp) + AlignmentStatistics.cpp, compiled as they
lie (oracle/Makefile), behind the glue of oracle/ref_stats.cpp."""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_PATH = os.path.join(_HERE, "_ref", "libpg_refstats.so")


def have_ref():
    return os.path.exists(REF_PATH)


def alignment_statistics(node_names, node_seqs, edges, edge_labels, reads):
    """edges: [(from, to)]; edge_labels: {(from, to): [label, ...]}; reads: [{pos, cigar, reverse, score, sequences: [..]}]
    (all MAPPED) -> the "alignment_statistics" object of the reference's count document."""
    L = C.CDLL(REF_PATH)
    L.pgrefs_alignment_statistics.restype = C.c_long

    def strs(items):
        return (C.c_char_p * max(1, len(items)))(*[s.encode() for s in items])

    def arr(items, dt):
        return np.asarray(list(items) or [0], dtype=dt)

    labels, label_off = [], [0]
    for e in edges:
        labels += list(edge_labels.get(tuple(e), []))
        label_off.append(len(labels))
    seqs, seq_off = [], [0]
    for r in reads:
        seqs += list(r["sequences"])
        seq_off.append(len(seqs))
    frm, to = arr((e[0] for e in edges), np.uint32), arr((e[1] for e in edges), np.uint32)
    loff, soff = arr(label_off, np.uint32), arr(seq_off, np.uint32)
    pos, score = arr((r["pos"] for r in reads), np.int32), arr((r["score"] for r in reads), np.int32)
    rev = arr((1 if r["reverse"] else 0 for r in reads), np.uint8)
    out = C.create_string_buffer(1 << 22)
    n = L.pgrefs_alignment_statistics(
        C.c_uint32(len(node_names)), strs(node_names), strs(node_seqs), C.c_uint32(len(edges)), frm.ctypes.data_as(C.c_void_p),
        to.ctypes.data_as(C.c_void_p), loff.ctypes.data_as(C.c_void_p), strs(labels), C.c_uint32(len(reads)),
        pos.ctypes.data_as(C.c_void_p), strs([r["cigar"] for r in reads]), rev.ctypes.data_as(C.c_void_p),
        score.ctypes.data_as(C.c_void_p), soff.ctypes.data_as(C.c_void_p), strs(seqs), out, C.c_size_t(len(out)))
    if n < 0:
        raise RuntimeError("pgrefs_alignment_statistics failed")
    return json.loads(out.value.decode())
