"""ctypes loaders for the two CPU checkers under oracle/ -- TEST INFRASTRUCTURE ONLY.

* ``RefOracle``  -> oracle/_ref/libpg_ref.so : the reference's own gssw.c driven by ref_harness.c
* ``PortOracle`` -> oracle/libpg_oracle.so   : plain-C restatement (pg_oracle.c)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing under paragraph_amd/ does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

AF_CIGAR = 1
AF_BOTH_STRANDS = 2
AF_REVERSE_GRAPH = 4
AF_ALL = 0xFFFFFFFF


class AlignResult(C.Structure):
    _fields_ = [
        ("graph_pos", C.c_int32),
        ("score", C.c_int32),
        ("mapq", C.c_int32),
        ("unique", C.c_int32),
        ("returned_reverse", C.c_int32),
        ("multi", C.c_int32 * 4),
        ("scores", C.c_int32 * 4),
        ("cigar_len", C.c_int32),
    ]


class FillResult(C.Structure):
    _fields_ = [
        ("score", C.c_int32),
        ("position", C.c_int32),
        ("max_node", C.c_int32),
        ("ref_end", C.c_int32),
        ("read_end", C.c_int32),
        ("multi", C.c_int32),
        ("cigar_len", C.c_int32),
    ]


def graph_csr(node_seqs, edges):
    """(node sequences, [(from,to)...]) -> CSR arrays (seq_off, seq, pred_off, pred); preds ascending."""
    n = len(node_seqs)
    seq_off = np.zeros(n + 1, dtype=np.uint32)
    for i, s in enumerate(node_seqs):
        seq_off[i + 1] = seq_off[i] + len(s)
    seq = "".join(node_seqs).encode("ascii")
    preds = [[] for _ in range(n)]
    for f, t in edges:
        if not (0 <= f < t < n):
            raise ValueError("edge (%d,%d) breaks topological order" % (f, t))
        preds[t].append(f)
    pred_off = np.zeros(n + 1, dtype=np.uint32)
    flat = []
    for i in range(n):
        ps = sorted(set(preds[i]))
        flat.extend(ps)
        pred_off[i + 1] = len(flat)
    pred = np.asarray(flat if flat else [0], dtype=np.uint32)
    return seq_off, seq, pred_off, pred


def pack_reads(reads):
    n = len(reads)
    off = np.zeros(n + 1, dtype=np.uint32)
    for i, r in enumerate(reads):
        off[i + 1] = off[i] + len(r)
    return off, "".join(reads).encode("ascii")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class _Base:
    prefix = None
    path = None

    def __init__(self):
        if not os.path.exists(self.path):
            raise FileNotFoundError(self.path)
        L = C.CDLL(self.path)
        pre = self.prefix
        u32p = C.POINTER(C.c_uint32)
        f = getattr(L, pre + "_graph_create")
        f.restype = C.c_void_p
        f.argtypes = [C.c_uint32, u32p, C.c_char_p, u32p, u32p]
        f = getattr(L, pre + "_graph_destroy")
        f.restype = None
        f.argtypes = [C.c_void_p]
        f = getattr(L, pre + "_align_read")
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_uint, C.POINTER(AlignResult), C.c_char_p, C.c_int]
        f = getattr(L, pre + "_fill")
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(FillResult), C.c_char_p, C.c_int]
        f = getattr(L, pre + "_align_batch")
        f.restype = C.c_int
        f.argtypes = [C.c_uint32, u32p, C.c_char_p, u32p, u32p, C.c_uint32, u32p, C.c_char_p, C.c_uint,
                      C.c_uint32, C.POINTER(AlignResult), C.c_char_p, C.c_int]
        self.L = L

    def _fn(self, name):
        return getattr(self.L, self.prefix + "_" + name)

    def graph(self, node_seqs, edges):
        return _Graph(self, node_seqs, edges)

    def align_batch(self, node_seqs, edges, reads, flags=AF_ALL, threads=1, cigar_stride=512, want_cigars=True):
        seq_off, seq, pred_off, pred = graph_csr(node_seqs, edges)
        off, bases = pack_reads(reads)
        n = len(reads)
        res = (AlignResult * max(n, 1))()
        cig = C.create_string_buffer(max(n, 1) * cigar_stride) if want_cigars else None
        u32 = C.c_uint32
        rc = self._fn("align_batch")(len(node_seqs), _p(seq_off, u32), seq, _p(pred_off, u32), _p(pred, u32), n,
                                     _p(off, u32), bases, flags & 0xFFFFFFFF, threads, res, cig, cigar_stride)
        if rc != 0:
            raise RuntimeError("%s_align_batch failed: %d" % (self.prefix, rc))
        out = []
        for i in range(n):
            r = res[i]
            c = cig.raw[i * cigar_stride:(i + 1) * cigar_stride].split(b"\0", 1)[0].decode() if want_cigars else ""
            out.append(result_dict(r, c))
        return out


# numpy mirror of AlignResult (13 x int32)
RESULT_NP = np.dtype([("graph_pos", "<i4"), ("score", "<i4"), ("mapq", "<i4"), ("unique", "<i4"), ("returned_reverse", "<i4"),
                      ("multi", "<i4", (4,)), ("scores", "<i4", (4,)), ("cigar_len", "<i4")])
assert RESULT_NP.itemsize == C.sizeof(AlignResult)


def result_dict(r, cigar):
    return {
        "graph_pos": r.graph_pos,
        "score": r.score,
        "mapq": r.mapq,
        "unique": bool(r.unique),
        "returned_reverse": bool(r.returned_reverse),
        "multi": [int(x) for x in r.multi],
        "scores": [int(x) for x in r.scores],
        "cigar": cigar,
    }


def _align_into(self, node_seqs, edges, off, bases, res, cig, threads=1, flags=AF_ALL):
    """Array form of align_batch for large samples: off = uint32 offsets[n+1], bases = uint8 array, res = RESULT_NP array
    (n entries), cig = (n, stride) uint8 array or None; both outputs may live in shared memory (they are written in place).
    Chunk-per-thread inside this process (threads), as Align.cpp:114-156."""
    seq_off, seq, pred_off, pred = graph_csr(node_seqs, edges)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    n = len(off) - 1
    if off[0] != 0:
        bases = bases[int(off[0]):int(off[-1])]
        off = off - off[0]
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    assert res.dtype == RESULT_NP and len(res) >= n and res.flags.c_contiguous
    stride = 0
    cig_p = None
    if cig is not None:
        assert cig.dtype == np.uint8 and cig.ndim == 2 and len(cig) >= n and cig.flags.c_contiguous
        stride = cig.shape[1]
        cig_p = C.cast(cig.ctypes.data, C.c_char_p)
    u32 = C.c_uint32
    rc = self._fn("align_batch")(len(node_seqs), _p(seq_off, u32), seq, _p(pred_off, u32), _p(pred, u32), n,
                                 _p(off, u32), C.cast(bases.ctypes.data, C.c_char_p), flags & 0xFFFFFFFF, threads,
                                 C.cast(res.ctypes.data, C.POINTER(AlignResult)), cig_p, stride)
    if rc != 0:
        raise RuntimeError("%s_align_batch failed: %d" % (self.prefix, rc))


_Base.align_into = _align_into


class _Graph:
    def __init__(self, lib, node_seqs, edges):
        self.lib = lib
        self.node_seqs = list(node_seqs)
        self.edges = list(edges)
        seq_off, seq, pred_off, pred = graph_csr(node_seqs, edges)
        self._keep = (seq_off, seq, pred_off, pred)
        u32 = C.c_uint32
        self.h = lib._fn("graph_create")(len(node_seqs), _p(seq_off, u32), seq, _p(pred_off, u32), _p(pred, u32))
        if not self.h:
            raise RuntimeError("graph_create failed")

    def close(self):
        if self.h:
            self.lib._fn("graph_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def align_read(self, bases, flags=AF_ALL):
        res = AlignResult()
        buf = C.create_string_buffer(4096)
        b = bases.encode("ascii")
        rc = self.lib._fn("align_read")(self.h, b, len(b), flags & 0xFFFFFFFF, C.byref(res), buf, len(buf))
        if rc != 0:
            raise RuntimeError("align_read failed: %d" % rc)
        return result_dict(res, buf.value.decode())

    def fill(self, direction, string):
        res = FillResult()
        buf = C.create_string_buffer(4096)
        b = string.encode("ascii")
        rc = self.lib._fn("fill")(self.h, direction, b, len(b), C.byref(res), buf, len(buf))
        if rc != 0:
            raise RuntimeError("fill failed: %d" % rc)
        return {
            "score": res.score, "position": res.position, "max_node": res.max_node, "ref_end": res.ref_end,
            "read_end": res.read_end, "multi": bool(res.multi), "cigar": buf.value.decode(),
        }


class RefOracle(_Base):
    """The reference's own gssw.c (oracle/_ref/libpg_ref.so)."""
    prefix = "pgref"
    path = os.path.join(_HERE, "_ref", "libpg_ref.so")

    def __init__(self):
        super().__init__()
        f = self.L.pgref_dump_matrices
        f.restype = C.c_int
        u8p = C.POINTER(C.c_uint8)
        f.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, u8p, u8p, u8p]

    def dump_matrices(self, graph, direction, node, read_len):
        n = len(graph.node_seqs[node if direction == 0 else len(graph.node_seqs) - 1 - node])
        H = np.zeros((n, read_len), dtype=np.uint8)
        E = np.zeros_like(H)
        F = np.zeros_like(H)
        rc = self.L.pgref_dump_matrices(graph.h, direction, node, read_len, _p(H, C.c_uint8), _p(E, C.c_uint8),
                                        _p(F, C.c_uint8))
        if rc != 0:
            raise RuntimeError("dump_matrices failed")
        return H, E, F


class PortOracle(_Base):
    """Plain-C restatement (oracle/libpg_oracle.so)."""
    prefix = "pgo"
    path = os.path.join(_HERE, "libpg_oracle.so")


def have_ref():
    return os.path.exists(RefOracle.path)


def have_port():
    return os.path.exists(PortOracle.path)
