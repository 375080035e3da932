"""The scalar core of the GENERAL gssw stage (paragraph_amd/csrc/pg_general.h: reads longer than 512 bp, graphs longer than
65 519 columns) on the CPU against the reference's own gssw.c: every pg_result field, the multi flags of all four fills and
the CIGAR.  The core is `__host__ __device__` code; tests/host_cpp/general_cpu.hip calls it on the host, the product calls
it from pg_general.hip's kernels (tests/test_gpu_general.py runs those)."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from tests import fuzzgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "host_cpp", "libpg_general_cpu.so")
SRC = os.path.join(ROOT, "tests", "host_cpp", "general_cpu.hip")
HDR = os.path.join(ROOT, "paragraph_amd", "csrc", "pg_general.h")


@pytest.fixture(scope="module")
def general():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared",
                        "-o", LIB, SRC], check=True)
    lib = C.CDLL(LIB)
    lib.pgt_general_align.restype = C.c_int
    from oracle import oracle as orc
    from paragraph_amd import capi

    def align(seqs, edges, read, flags=0xFFFFFFFF):
        seq_off, seq, pred_off, pred = orc.graph_csr(seqs, edges)
        res = np.zeros(1, dtype=capi.RESULT_DTYPE)
        ops = np.zeros(len(read) + 64, dtype=np.uint32)
        n_ops = C.c_uint32(0)
        fills = np.zeros((4, 6), dtype=np.int32)
        b = read.encode("ascii")
        rc = lib.pgt_general_align(len(seqs), seq_off.ctypes.data_as(C.c_void_p), seq, pred_off.ctypes.data_as(C.c_void_p),
                                   pred.ctypes.data_as(C.c_void_p), b, len(b), C.c_uint32(flags), res.ctypes.data_as(C.c_void_p),
                                   ops.ctypes.data_as(C.c_void_p), len(ops), C.byref(n_ops), fills.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return capi.results_to_dicts(res, ops)[0], fills

    return align


def _same(got, want):
    if want["score"] == 0:  # degenerate: the reference's behaviour downstream is undefined; score 0, no CIGAR, status 1
        return got["score"] == 0 and got["status"] == 1 and got["cigar"] == "" and got["multi"] == list(want["multi"]) \
            and got["unique"] == bool(want["unique"])
    return all(got[k] == want[k] for k in ("graph_pos", "score", "mapq", "cigar")) and got["unique"] == bool(want["unique"]) \
        and got["returned_reverse"] == bool(want["returned_reverse"]) and got["multi"] == list(want["multi"]) and got["status"] == 0


def test_general_core_short_and_medium_reads(general, checker):
    rng = random.Random(4242)
    n = 0
    for it in range(260):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=rng.choice([6, 40, 120]), max_nodes=8)
        reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=1, max_len=rng.choice([12, 60, 200])) for _ in range(6)]
        for r, w in zip(reads, checker.align_batch(seqs, edges, reads)):
            g, _ = general(seqs, edges, r)
            assert _same(g, w), (seqs, edges, r, g, w)
            n += 1
    assert n == 1560


def test_general_core_long_reads_word_mode(general, checker):
    """reads of 251..1200 bases: scores beyond gssw's byte mode, incl. alignsEndAtMultNodes' byte view of the 16-bit matrix"""
    rng = random.Random(777)
    n = n_over = n_edge = n_long = 0
    for it in range(90):
        if it % 3 == 0:
            seqs, edges = fuzzgen.rand_graph(rng, max_len=rng.choice([260, 700]), max_nodes=5)
            reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=251, max_len=1200) for _ in range(4)]
        elif it % 3 == 2:  # certainly longer than the packed kernels' 512 bases
            seqs = [fuzzgen.rand_seq(rng, rng.randint(300, 900)) for _ in range(4)]
            edges = [(0, 1), (0, 2), (1, 2), (1, 3), (2, 3)]
            reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=520, max_len=1200) for _ in range(4)]
        else:
            seqs, edges, reads = fuzzgen.long_read_case(rng, n_reads=4)
        for r, w in zip(reads, checker.align_batch(seqs, edges, reads, cigar_stride=4096)):
            g, _ = general(seqs, edges, r)
            assert _same(g, w), (seqs, edges, r, g, w)
            n += 1
            n_over += w["score"] >= 251
            n_edge += 251 <= max(w["scores"]) <= 255
            n_long += len(r) > 512
    assert n == 360 and n_over > 100 and n_edge > 5 and n_long > 40, (n, n_over, n_edge, n_long)


def test_general_core_flag_subsets(general, checker):
    rng = random.Random(99)
    for it in range(40):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=50, max_nodes=6)
        r = fuzzgen.rand_read(rng, seqs, edges, min_len=5, max_len=90)
        for flags in (1, 3, 5, 7):
            w = checker.align_batch(seqs, edges, [r], flags=flags)[0]
            g, _ = general(seqs, edges, r, flags)
            assert _same(g, w), (flags, seqs, edges, r, g, w)
