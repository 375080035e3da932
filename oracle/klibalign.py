"""KlibAligner checkers -- TEST INFRASTRUCTURE ONLY.

* ``ref_*``  : oracle/_ref/libpg_ref.so = the reference's own external/klib/ksw.c (ksw_align + ksw_global) under the
               restated wrapper of oracle/klib_glue.h (KlibAlignment::update, KlibAligner::alignRead / pickBest /
               buildGraphCigar -- src/c++/lib/common/Klib.cpp:144-164, src/c++/lib/grm/KlibAligner.cpp:186-442)
* ``port_*`` : oracle/libpg_oracle.so = scalar restatement of ksw.c (pg_oracle.c) under the same wrapper

Both are pinned on src/c++/test/test_klibaligner.cpp:149-193 and src/c++/test/test_align.cpp:38-263
(tests/test_klib_oracle.py) and against each other on randomized inputs.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(_HERE, "_ref", "libpg_ref.so")
PORT_LIB = os.path.join(_HERE, "libpg_oracle.so")

_OPS = "MIDNS"


class KlibResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("status", "graph_pos", "score", "mapq", "unique", "is_graph_reverse", "used_reverse", "ub",
                                         "cigar_len")]


def have_ref():
    return os.path.exists(REF_LIB)


class _Klib:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        u32p = C.POINTER(C.c_uint32)
        self._pair = getattr(self.lib, prefix + "_klib_pair")
        self._pair.restype = C.c_int
        self._pair.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), u32p, C.c_int]
        self._align = getattr(self.lib, prefix + "_klib_align")
        self._align.restype = C.c_int
        self._align.argtypes = [C.c_int, u32p, C.c_char_p, C.c_int, u32p, u32p, C.c_uint32, u32p, C.c_char_p, C.c_char_p,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(KlibResult), C.c_char_p, C.c_int]

    def pair(self, ref, query, match=2, mismatch=-2, gapo=3, gape=1):
        """One common::KlibAlignment (default parameters = AlignmentParameters()). Returns dict or None (undefined case)."""
        out5 = (C.c_int32 * 5)()
        cap = 2 * (len(ref) + len(query)) + 8
        cig = (C.c_uint32 * cap)()
        n = self._pair(ref.encode(), query.encode(), match, mismatch, gapo, gape, out5, cig, cap)
        if n < 0:
            return None
        cigar = "".join("%d%s" % (cig[i] >> 4, _OPS[cig[i] & 0xF]) for i in range(n))
        return {"score": out5[0], "r0": out5[1], "r1": out5[2], "a0": out5[3], "a1": out5[4], "cigar": cigar}

    def align(self, nodes, paths, reads, bam_reverse=None, match=1, mismatch=-4, gapo=5, gape=1):
        u32p = C.POINTER(C.c_uint32)
        node_off = np.zeros(len(nodes) + 1, dtype=np.uint32)
        node_off[1:] = np.cumsum([len(s) for s in nodes])
        pno = np.zeros(len(paths) + 1, dtype=np.uint32)
        pno[1:] = np.cumsum([len(p) for p in paths])
        pn = np.array([x for p in paths for x in p] or [0], dtype=np.uint32)
        roff = np.zeros(len(reads) + 1, dtype=np.uint32)
        roff[1:] = np.cumsum([len(r) for r in reads])
        res = (KlibResult * max(1, len(reads)))()
        stride = 4096
        cig = C.create_string_buffer(max(1, len(reads)) * stride)
        br = None
        if bam_reverse is not None:
            br = bytes(bytearray(int(bool(x)) for x in bam_reverse))

        def P(a):
            return a.ctypes.data_as(u32p)
        self._align(len(nodes), P(node_off), "".join(nodes).encode(), len(paths), P(pno), P(pn), len(reads), P(roff),
                    "".join(reads).encode(), br, match, mismatch, gapo, gape, res, cig, stride)
        out = []
        for i in range(len(reads)):
            r = res[i]
            out.append({"status": r.status, "graph_pos": r.graph_pos, "score": r.score, "mapq": r.mapq, "unique": bool(r.unique),
                        "is_graph_reverse": bool(r.is_graph_reverse), "used_reverse": bool(r.used_reverse), "ub": bool(r.ub),
                        "cigar": cig.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode()})
        return out


_cache = {}


def ref_klib():
    if "ref" not in _cache:
        _cache["ref"] = _Klib(REF_LIB, "pgref")
    return _cache["ref"]


def port_klib():
    if "port" not in _cache:
        _cache["port"] = _Klib(PORT_LIB, "pgo")
    return _cache["port"]
