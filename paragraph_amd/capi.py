"""ctypes binding of the C ABI in include/paragraph_amd.h (libparagraph_amd.so, HIP/gfx950).

This module is plumbing only: it never computes an alignment itself and has no CPU fallback --
if the shared library is missing or no HIP device is usable, it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PG_LIB") or os.path.join(_HERE, "libparagraph_amd.so")  # PG_LIB: another build of the same ABI (tools/build_variant.sh, A/B timing)

AF_CIGAR = 1
AF_BOTH_STRANDS = 2
AF_REVERSE_GRAPH = 4
AF_ALL = 0xFFFFFFFF
AF_KEEP_RESULTS = 0x100
STATUS_PATH_ALIGNER = 0x100
STATUS_KMER_ALIGNER = 0x200
STATUS_KLIB_ALIGNER = 0x400

PG_OK = 0
STATUS_NAMES = {0: "PG_OK", 1: "PG_ERR_INVALID", 2: "PG_ERR_NO_DEVICE", 3: "PG_ERR_HIP", 4: "PG_ERR_UNSUPPORTED",
                5: "PG_ERR_NOMEM", 6: "PG_ERR_OVERFLOW"}

# numpy mirror of struct pg_result (24 bytes)
RESULT_DTYPE = np.dtype([
    ("graph_pos", "<i4"), ("score", "<i2"), ("mapq", "u1"), ("is_unique", "u1"), ("returned_reverse", "u1"),
    ("multi_mask", "u1"), ("n_ops", "<u2"), ("ops_off", "<u4"), ("strand_score", "<i2", (2,)), ("clipped", "<u2"),
    ("status", "<u2"),
], align=True)
assert RESULT_DTYPE.itemsize == 24, RESULT_DTYPE.itemsize

OP_CHARS = "MXNIDS"

SUPPORT_DTYPE = np.dtype([("label_mask", "<u8"), ("path_off", "<u4"), ("n_path", "<u2"), ("status", "u1"),
                          ("filter", "u1")], align=True)
assert SUPPORT_DTYPE.itemsize == 16, SUPPORT_DTYPE.itemsize

EXPORTS = [
    "pg_device_prefer_blocking_waits", "pg_ctx_create", "pg_ctx_destroy", "pg_strerror", "pg_last_error", "pg_ctx_set_workspace_bytes", "pg_ctx_sync",
    "pg_ctx_timing_enable", "pg_ctx_timing_reset", "pg_ctx_timing_get", "pg_graphs_upload", "pg_graphs_destroy",
    "pg_batch_create", "pg_batch_destroy", "pg_batch_upload", "pg_batch_align", "pg_batch_ops_count",
    "pg_batch_download", "pg_align_batch", "pg_render_cigar", "pg_graphs_set_labels", "pg_graphs_set_labels_wide", "pg_graphs_label_words",
    "pg_batch_download_label_ext", "pg_graphs_count_layout",
    "pg_graphs_seq_offsets", "pg_batch_set_fragments", "pg_batch_count", "pg_batch_download_counts", "pg_graphs_build_path_index",
    "pg_batch_path_align", "pg_batch_download_path_flags", "pg_batch_set_active", "pg_graphs_build_kmer_index",
    "pg_batch_kmer_align", "pg_graphs_build_klib_index", "pg_batch_klib_align", "pg_graphs_klib_error", "pg_graphs_klib_last_kernels", "pg_graphs_build_filter_index",
    "pg_host_alloc", "pg_host_free", "pg_host_register", "pg_host_unregister", "pg_counts_zero", "pg_ctx_sync_compute",
    "pg_render_cigars", "pg_ctx_native_stream", "pg_ctx_count_record", "pg_ctx_count_wait", "pg_ctx_set_fill_streams", "pg_ctx_set_lean",
    "pg_batch_retire_mapped", "pg_batch_result_sizes", "pg_batch_download_all", "pg_batch_retire_exact_matches",
]


class Timing(C.Structure):
    _fields_ = [("fill_ms", C.c_double), ("trace_ms", C.c_double), ("fill_launches", C.c_uint64),
                ("trace_launches", C.c_uint64), ("fills", C.c_uint64), ("cells", C.c_uint64),
                ("trace_bytes", C.c_uint64), ("lean_rev_ms", C.c_double), ("lean_fwd_ms", C.c_double),
                ("lean_rev_launches", C.c_uint64), ("lean_fwd_launches", C.c_uint64), ("lean_fused_ms", C.c_double),
                ("lean_fused_launches", C.c_uint64)]


class CountParams(C.Structure):
    _fields_ = [("remove_nonuniq", C.c_uint32), ("use_support_filters", C.c_uint32), ("bad_align_frac", C.c_double),
                ("use_kmer_filter", C.c_uint32), ("reserved", C.c_uint32)]


class CountLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_counters", "node_base", "edge_base", "seq_base", "tally_base", "n_nodes",
                                           "n_edges", "n_seq_slots", "n_graphs")]


class PgError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("%s (%d)%s" % (STATUS_NAMES.get(status, "?"), status, ": " + msg if msg else ""))


_lib = None


def load_library():
    """Loads libparagraph_amd.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    u32p = C.POINTER(C.c_uint32)
    vp = C.c_void_p
    L.pg_device_prefer_blocking_waits.restype = C.c_int32
    L.pg_device_prefer_blocking_waits.argtypes = [C.c_int]
    L.pg_ctx_create.restype = C.c_int32
    L.pg_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.pg_ctx_destroy.restype = None
    L.pg_ctx_destroy.argtypes = [vp]
    L.pg_strerror.restype = C.c_char_p
    L.pg_strerror.argtypes = [C.c_int32]
    L.pg_last_error.restype = C.c_char_p
    L.pg_last_error.argtypes = [vp]
    L.pg_ctx_set_workspace_bytes.restype = C.c_int32
    L.pg_ctx_set_workspace_bytes.argtypes = [vp, C.c_uint64]
    L.pg_ctx_set_fill_streams.restype = C.c_int32
    L.pg_ctx_set_fill_streams.argtypes = [vp, C.c_int]
    L.pg_ctx_set_lean.restype = C.c_int32
    L.pg_ctx_set_lean.argtypes = [vp, C.c_int]
    L.pg_ctx_sync.restype = C.c_int32
    L.pg_ctx_sync.argtypes = [vp]
    L.pg_ctx_timing_enable.restype = C.c_int32
    L.pg_ctx_timing_enable.argtypes = [vp, C.c_int]
    L.pg_ctx_timing_reset.restype = C.c_int32
    L.pg_ctx_timing_reset.argtypes = [vp]
    L.pg_ctx_timing_get.restype = C.c_int32
    L.pg_ctx_timing_get.argtypes = [vp, C.POINTER(Timing)]
    L.pg_graphs_upload.restype = C.c_int32
    L.pg_graphs_upload.argtypes = [vp, C.c_uint32, u32p, u32p, C.c_char_p, u32p, u32p, C.POINTER(vp)]
    L.pg_graphs_destroy.restype = None
    L.pg_graphs_destroy.argtypes = [vp, vp]
    L.pg_batch_create.restype = C.c_int32
    L.pg_batch_create.argtypes = [vp, C.POINTER(vp)]
    L.pg_batch_destroy.restype = None
    L.pg_batch_destroy.argtypes = [vp, vp]
    L.pg_batch_upload.restype = C.c_int32
    L.pg_batch_upload.argtypes = [vp, vp, vp, C.c_uint32, u32p, u32p, vp]
    L.pg_batch_align.restype = C.c_int32
    L.pg_batch_align.argtypes = [vp, vp, C.c_uint32]
    L.pg_batch_ops_count.restype = C.c_int32
    L.pg_batch_ops_count.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    L.pg_batch_download.restype = C.c_int32
    L.pg_batch_download.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.pg_align_batch.restype = C.c_int32
    L.pg_align_batch.argtypes = [vp, vp, C.c_uint32, u32p, u32p, C.c_char_p, C.c_uint32, vp, vp, C.c_uint64,
                                 C.POINTER(C.c_uint64)]
    u64p = C.POINTER(C.c_uint64)
    L.pg_graphs_set_labels.restype = C.c_int32
    L.pg_graphs_set_labels.argtypes = [vp, vp, u64p, u32p]
    L.pg_graphs_set_labels_wide.restype = C.c_int32
    L.pg_graphs_set_labels_wide.argtypes = [vp, vp, u64p, C.c_uint32, u32p]
    L.pg_graphs_label_words.restype = C.c_int32
    L.pg_graphs_label_words.argtypes = [vp, u32p]
    L.pg_batch_download_label_ext.restype = C.c_int32
    L.pg_batch_download_label_ext.argtypes = [vp, vp, u64p, C.c_uint64]
    L.pg_graphs_count_layout.restype = C.c_int32
    L.pg_graphs_count_layout.argtypes = [vp, C.POINTER(CountLayout)]
    L.pg_graphs_seq_offsets.restype = C.c_int32
    L.pg_graphs_seq_offsets.argtypes = [vp, u64p]
    L.pg_batch_count.restype = C.c_int32
    L.pg_batch_count.argtypes = [vp, vp, C.POINTER(CountParams), vp]
    L.pg_batch_set_fragments.restype = C.c_int32
    L.pg_batch_set_fragments.argtypes = [vp, vp, u32p, C.POINTER(C.c_uint8)]
    L.pg_batch_download_counts.restype = C.c_int32
    L.pg_batch_download_counts.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, u64p]
    L.pg_graphs_build_path_index.restype = C.c_int32
    L.pg_graphs_build_path_index.argtypes = [vp, vp, C.c_uint32]
    L.pg_batch_path_align.restype = C.c_int32
    L.pg_batch_path_align.argtypes = [vp, vp]
    L.pg_batch_download_path_flags.restype = C.c_int32
    L.pg_batch_download_path_flags.argtypes = [vp, vp, vp]
    L.pg_batch_set_active.restype = C.c_int32
    L.pg_batch_set_active.argtypes = [vp, vp, vp]
    L.pg_batch_retire_mapped.restype = C.c_int32
    L.pg_batch_retire_mapped.argtypes = [vp, vp]
    L.pg_batch_retire_exact_matches.restype = C.c_int32
    L.pg_batch_retire_exact_matches.argtypes = [vp, vp]
    L.pg_batch_result_sizes.restype = C.c_int32
    L.pg_batch_result_sizes.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pg_batch_download_all.restype = C.c_int32
    L.pg_batch_download_all.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64]
    L.pg_graphs_build_kmer_index.restype = C.c_int32
    L.pg_graphs_build_kmer_index.argtypes = [vp, vp, C.c_uint32, u32p, u32p, u32p]
    L.pg_batch_kmer_align.restype = C.c_int32
    L.pg_batch_kmer_align.argtypes = [vp, vp, C.c_uint32]
    L.pg_graphs_build_filter_index.restype = C.c_int32
    L.pg_graphs_build_filter_index.argtypes = [vp, vp, C.c_int32, u32p]
    L.pg_graphs_build_klib_index.restype = C.c_int32
    L.pg_graphs_build_klib_index.argtypes = [vp, vp, u32p, u32p, u32p]
    L.pg_batch_klib_align.restype = C.c_int32
    L.pg_batch_klib_align.argtypes = [vp, vp, C.c_uint32]
    L.pg_graphs_klib_error.restype = C.c_int32
    L.pg_graphs_klib_error.argtypes = [vp, vp, u32p]
    L.pg_graphs_klib_last_kernels.restype = C.c_int32
    L.pg_graphs_klib_last_kernels.argtypes = [vp, vp, u32p]
    L.pg_host_alloc.restype = C.c_int32
    L.pg_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.pg_host_free.restype = None
    L.pg_host_free.argtypes = [vp, vp]
    L.pg_host_register.restype = C.c_int32
    L.pg_host_register.argtypes = [vp, vp, C.c_size_t]
    L.pg_host_unregister.restype = C.c_int32
    L.pg_host_unregister.argtypes = [vp, vp]
    L.pg_counts_zero.restype = C.c_int32
    L.pg_counts_zero.argtypes = [vp, vp, C.c_uint64]
    L.pg_ctx_sync_compute.restype = C.c_int32
    L.pg_ctx_sync_compute.argtypes = [vp]
    L.pg_ctx_native_stream.restype = C.c_int32
    L.pg_ctx_native_stream.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.pg_ctx_count_record.restype = C.c_int32
    L.pg_ctx_count_record.argtypes = [vp, vp]
    L.pg_ctx_count_wait.restype = C.c_int32
    L.pg_ctx_count_wait.argtypes = [vp, vp]
    L.pg_render_cigars.restype = C.c_int32
    L.pg_render_cigars.argtypes = [vp, C.c_uint64, vp, vp, C.c_size_t]
    L.pg_render_cigar.restype = C.c_size_t
    L.pg_render_cigar.argtypes = [vp, vp, C.c_char_p, C.c_size_t]
    _lib = L
    return L


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def graphs_csr(graphs):
    """graphs: list of (node_seqs, edges) -> CSR arrays for pg_graphs_upload."""
    node_off = [0]
    seq_off = [0]
    seqs = []
    pred_off = [0]
    pred = []
    for node_seqs, edges in graphs:
        n = len(node_seqs)
        preds = [[] for _ in range(n)]
        for f, t in edges:
            if not (0 <= f < t < n):
                raise ValueError("edge (%d,%d) breaks topological order" % (f, t))
            preds[t].append(f)
        for i, s in enumerate(node_seqs):
            seqs.append(s)
            seq_off.append(seq_off[-1] + len(s))
            ps = sorted(set(preds[i]))
            pred.extend(ps)
            pred_off.append(len(pred))
        node_off.append(node_off[-1] + n)
    return (_u32(node_off), _u32(seq_off), "".join(seqs).encode("ascii"), _u32(pred_off),
            _u32(pred if pred else [0]))


def _bases_ptr(bases):
    """bytes or a uint8 numpy array (e.g. over pinned memory) -> (address, keep-alive object)."""
    if isinstance(bases, np.ndarray):
        a = np.ascontiguousarray(bases, dtype=np.uint8)
        return a.ctypes.data, a
    buf = C.c_char_p(bases)
    return C.cast(buf, C.c_void_p).value, buf


class _PinnedBlock:
    """Owner of one pg_host_alloc block.  The ctypes buffer numpy arrays are made over carries a reference to its block
    (`_owner`), so the block is released when the last array / view over it goes away -- not when the context does."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        p = C.c_void_p()
        ctx._chk(ctx.L.pg_host_alloc(ctx.h, max(int(nbytes), 1), C.byref(p)))
        self.ptr = p
        self.nbytes = max(int(nbytes), 1)

    def buffer(self):
        buf = (C.c_uint8 * self.nbytes).from_address(self.ptr.value)
        buf._owner = self  # buffer -> block; the block holds no reference back (no cycle)
        return buf

    def __del__(self):
        try:
            if self.ptr and self.ctx.h:
                self.ctx.L.pg_host_free(self.ctx.h, self.ptr)
            self.ptr = None
        except Exception:
            pass


def pack_reads(reads):
    """list[str] or (offsets, bytes | uint8 array) -> (uint32 offsets[n+1], bytes | uint8 array)."""
    if isinstance(reads, tuple):
        return _u32(reads[0]), reads[1]
    lens = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
    off = np.zeros(len(reads) + 1, dtype=np.uint32)
    np.cumsum(lens, out=off[1:])
    return off, "".join(reads).encode("ascii")


class Context:
    """One pg_ctx (one device, one stream)."""

    def __init__(self, device=0, workspace_bytes=None, fill_streams=None):
        self.L = load_library()
        h = C.c_void_p()
        st = self.L.pg_ctx_create(device, C.byref(h))
        if st != PG_OK:
            raise PgError(st, "pg_ctx_create(device=%d)" % device)
        self.h = h
        if workspace_bytes:
            self._chk(self.L.pg_ctx_set_workspace_bytes(self.h, workspace_bytes))
        if fill_streams:
            self.set_fill_streams(fill_streams)

    def set_lean(self, on):
        """the lean gssw stage (include/paragraph_amd.h, pg_ctx_set_lean): alignRead(AF_ALL) from three fills per read where the fourth
        cannot change the record"""
        self._chk(self.L.pg_ctx_set_lean(self.h, int(on)))  # (0 off, 1 on for chunks of 30 G cell updates and more, 2 on for every chunk)

    def set_fill_streams(self, n):
        """1: fills one after the other on the main stream (two workspace regions); 2: fills alternate over two streams (three
        regions) -- short launches then overlap their tails.  Before batches are uploaded; results do not depend on it."""
        self._chk(self.L.pg_ctx_set_fill_streams(self.h, n))

    def _chk(self, st):
        if st != PG_OK:
            raise PgError(st, (self.L.pg_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            self.L.pg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self.L.pg_ctx_sync(self.h))

    def sync_compute(self):
        self._chk(self.L.pg_ctx_sync_compute(self.h))

    def pinned_empty(self, shape, dtype):
        """numpy array over page-locked host memory (pg_host_alloc); the block is freed when the last array / view over it
        goes away (it must not outlive the context)."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if not isinstance(shape, int) else int(shape)
        blk = _PinnedBlock(self, n * dt.itemsize)
        a = np.frombuffer(blk.buffer(), dtype=dt, count=n)
        return a.reshape(shape)

    def pinned_copy(self, arr):
        a = np.ascontiguousarray(arr)
        out = self.pinned_empty(a.shape, a.dtype)
        out[...] = a
        return out

    def native_stream(self, which=1):
        """hipStream_t of the ctx as an integer: 0 fill stream, 1 count stream (traceback + count path), 2 copy stream."""
        p = C.c_void_p()
        self._chk(self.L.pg_ctx_native_stream(self.h, int(which), C.byref(p)))
        return p.value

    def count_record(self, native_event):
        """hipEventRecord(event, count stream) -- `native_event` = the hipEvent_t as an integer."""
        self._chk(self.L.pg_ctx_count_record(self.h, C.c_void_p(int(native_event))))

    def count_wait(self, native_event):
        """hipStreamWaitEvent(count stream, event): later counts_zero / Batch.count calls start after the event."""
        self._chk(self.L.pg_ctx_count_wait(self.h, C.c_void_p(int(native_event))))

    def counts_zero(self, d_ptr, n_counters):
        """memset of a caller-owned device counter table on the ctx stream (ordered with Batch.count)."""
        self._chk(self.L.pg_counts_zero(self.h, d_ptr, int(n_counters)))

    def timing_enable(self, on=True):
        self._chk(self.L.pg_ctx_timing_enable(self.h, 1 if on else 0))

    def timing_reset(self):
        self._chk(self.L.pg_ctx_timing_reset(self.h))

    def timing(self):
        t = Timing()
        self._chk(self.L.pg_ctx_timing_get(self.h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in Timing._fields_}

    def upload_graphs(self, graphs):
        return Graphs(self, graphs)

    def new_batch(self):
        return Batch(self)


class Graphs:
    def __init__(self, ctx, graphs):
        self.ctx = ctx
        self.n = len(graphs)
        node_off, seq_off, seq, pred_off, pred = graphs_csr(graphs)
        self.node_off, self.pred_off, self.pred = node_off, pred_off, pred
        self.edges = [[(int(pred[k]), t) for t in range(int(node_off[g + 1] - node_off[g]))
                       for k in range(int(pred_off[node_off[g] + t]), int(pred_off[node_off[g] + t + 1]))]
                      for g in range(self.n)]
        self.labels = None
        h = C.c_void_p()
        ctx._chk(ctx.L.pg_graphs_upload(ctx.h, self.n, _p32(node_off), _p32(seq_off), seq, _p32(pred_off),
                                        _p32(pred), C.byref(h)))
        self.h = h

    def build_path_index(self, kmer_len=32):
        self.ctx._chk(self.ctx.L.pg_graphs_build_path_index(self.ctx.h, self.h, kmer_len))

    def build_kmer_index(self, paths, kmer_len=16):
        """paths: per graph a list of node-id lists (whole-node paths as grm::pathsFromJson builds them)."""
        poff, noff, nodes = [0], [0], []
        for gp in paths:
            for p in gp:
                nodes.extend(p)
                noff.append(len(nodes))
            poff.append(len(noff) - 1)
        poff, noff, nodes = _u32(poff), _u32(noff), _u32(nodes if nodes else [0])
        self.ctx._chk(self.ctx.L.pg_graphs_build_kmer_index(self.ctx.h, self.h, kmer_len, _p32(poff), _p32(noff), _p32(nodes)))

    def build_filter_index(self, kmer_len):
        """KmerFilter index; kmer_len < 0 = auto-detect per graph. Returns the lengths used."""
        out = np.zeros(max(self.n, 1), dtype=np.uint32)
        self.ctx._chk(self.ctx.L.pg_graphs_build_filter_index(self.ctx.h, self.h, int(kmer_len), _p32(out)))
        return [int(x) for x in out[:self.n]]

    def build_klib_index(self, paths):
        """paths: per graph a list of node-id lists (as for build_kmer_index)."""
        poff, noff, nodes = [0], [0], []
        for gp in paths:
            for p in gp:
                nodes.extend(p)
                noff.append(len(nodes))
            poff.append(len(noff) - 1)
        poff, noff, nodes = _u32(poff), _u32(noff), _u32(nodes if nodes else [0])
        self.ctx._chk(self.ctx.L.pg_graphs_build_klib_index(self.ctx.h, self.h, _p32(poff), _p32(noff), _p32(nodes)))

    def klib_error(self):
        e = np.zeros(1, dtype=np.uint32)
        self.ctx._chk(self.ctx.L.pg_graphs_klib_error(self.ctx.h, self.h, _p32(e)))
        return int(e[0])

    def klib_used_packed_kernels(self):
        """True when the last klib_align on this graph set ran the packed two-strand kernels (tests)."""
        e = np.zeros(1, dtype=np.uint32)
        self.ctx._chk(self.ctx.L.pg_graphs_klib_last_kernels(self.ctx.h, self.h, _p32(e)))
        return bool(e[0])

    def set_labels(self, edge_labels, labels=None):
        """edge_labels: per graph a dict {(from,to): [label,...]}; labels: per graph the ordered label list
        (default: sorted names).  Counters of edges come back in predecessor-CSR order = self.edges[g]."""
        if labels is None:
            labels = [sorted({l for v in el.values() for l in v}) for el in edge_labels]
        self.labels = [list(l) for l in labels]
        # label sets are bit sets of `words` 64-bit words for the whole graph set (1 unless a graph has more than 64 labels)
        self.label_words = words = max(1, max((len(l) + 63) // 64 for l in self.labels) if self.labels else 1)
        mask = np.zeros((max(1, len(self.pred)), words), dtype=np.uint64)
        k = 0
        for g in range(self.n):
            idx = {l: i for i, l in enumerate(self.labels[g])}
            for e in self.edges[g]:
                m = 0
                for l in edge_labels[g].get(e, []):
                    m |= 1 << idx[l]
                for w in range(words):
                    mask[k, w] = (m >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
                k += 1
        nl = _u32([len(l) for l in self.labels])
        if words == 1:
            self.ctx._chk(self.ctx.L.pg_graphs_set_labels(self.ctx.h, self.h, mask.ctypes.data_as(C.POINTER(C.c_uint64)), _p32(nl)))
        else:
            self.ctx._chk(self.ctx.L.pg_graphs_set_labels_wide(self.ctx.h, self.h, mask.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                           words, _p32(nl)))
        lay = CountLayout()
        self.ctx._chk(self.ctx.L.pg_graphs_count_layout(self.h, C.byref(lay)))
        self.layout = lay
        so = np.zeros(self.n + 1, dtype=np.uint64)
        self.ctx._chk(self.ctx.L.pg_graphs_seq_offsets(self.h, so.ctypes.data_as(C.POINTER(C.c_uint64))))
        self.seq_off = so

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.L.pg_graphs_destroy(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._chk(ctx.L.pg_batch_create(ctx.h, C.byref(h)))
        self.h = h
        self.n_reads = 0

    def close(self):
        if self.h and self.ctx.h:
            self.ctx.L.pg_batch_destroy(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, graphs, reads, graph_of_read=None):
        off, bases = pack_reads(reads)
        n = len(off) - 1
        gor = _u32(np.zeros(n, dtype=np.uint32) if graph_of_read is None else graph_of_read)  # no copy if already uint32
        if len(gor) != n:
            raise ValueError("graph_of_read length mismatch")
        self._graphs = graphs
        ptr, keep = _bases_ptr(bases)
        self.ctx._chk(self.ctx.L.pg_batch_upload(self.ctx.h, self.h, graphs.h, n, _p32(gor), _p32(off), ptr))
        del keep
        self.n_reads = n

    def path_align(self, fetch_flags=True):
        """PathAligner stage for every read; returns flags (bit0 mapped, bit1 anchored) -- or None with fetch_flags=False: the
        stage is then only queued (the device-resident cascade never reads the flags on the host)."""
        self.ctx._chk(self.ctx.L.pg_batch_path_align(self.ctx.h, self.h))
        if not fetch_flags:
            return None
        fl = np.zeros(max(self.n_reads, 1), dtype=np.uint8)
        self.ctx._chk(self.ctx.L.pg_batch_download_path_flags(self.ctx.h, self.h, fl.ctypes.data))
        return fl[:self.n_reads]

    def klib_align(self, flags=AF_ALL, fetch_flags=True):
        """KlibAligner stage for every active read; returns flags (bit0 mapped, bit2 BAD_ALIGN), or None with fetch_flags=False."""
        self.ctx._chk(self.ctx.L.pg_batch_klib_align(self.ctx.h, self.h, flags & 0xFFFFFFFF))
        if not fetch_flags:
            return None
        fl = np.zeros(max(self.n_reads, 1), dtype=np.uint8)
        self.ctx._chk(self.ctx.L.pg_batch_download_path_flags(self.ctx.h, self.h, fl.ctypes.data))
        return fl[:self.n_reads]

    def kmer_align(self, flags=AF_ALL, fetch_flags=True):
        """KmerAligner stage for every active read; returns flags (bit0 mapped, bit2 BAD_ALIGN), or None with fetch_flags=False."""
        self.ctx._chk(self.ctx.L.pg_batch_kmer_align(self.ctx.h, self.h, flags & 0xFFFFFFFF))
        if not fetch_flags:
            return None
        fl = np.zeros(max(self.n_reads, 1), dtype=np.uint8)
        self.ctx._chk(self.ctx.L.pg_batch_download_path_flags(self.ctx.h, self.h, fl.ctypes.data))
        return fl[:self.n_reads]

    def set_active(self, active):
        if active is None:
            self.ctx._chk(self.ctx.L.pg_batch_set_active(self.ctx.h, self.h, None))
        else:
            a = np.ascontiguousarray(active, dtype=np.uint8)
            if len(a) != self.n_reads:
                raise ValueError("active length mismatch")
            self.ctx._chk(self.ctx.L.pg_batch_set_active(self.ctx.h, self.h, a.ctypes.data))

    def retire_mapped(self):
        """The cascade hand-over decided on the device: after a seed stage + count(), reads the stage mapped and the filter chain
        accepted leave the following stages (pg_batch_retire_mapped); nothing is downloaded."""
        self.ctx._chk(self.ctx.L.pg_batch_retire_mapped(self.ctx.h, self.h))

    def retire_exact_matches(self):
        """After path_align(): the reads whose gssw record the path stage's one exact full-length match forces keep that record and
        leave the following align() (pg_batch_retire_exact_matches); nothing is downloaded."""
        self.ctx._chk(self.ctx.L.pg_batch_retire_exact_matches(self.ctx.h, self.h))

    def download_all(self, want_table=True):
        """-> (results, ops, counts table or None, supports, path entries) with one wait for the batch and one for the copies
        (pg_batch_result_sizes + pg_batch_download_all); the batch's last stage must be count()."""
        n_ops, n_path = C.c_uint64(), C.c_uint64()
        self.ctx._chk(self.ctx.L.pg_batch_result_sizes(self.ctx.h, self.h, C.byref(n_ops), C.byref(n_path)))
        res = np.zeros(max(self.n_reads, 1), dtype=RESULT_DTYPE)
        sup = np.zeros(max(self.n_reads, 1), dtype=SUPPORT_DTYPE)
        ops = np.zeros(max(int(n_ops.value), 1), dtype=np.uint32)
        path = np.zeros(max(int(n_path.value), 1), dtype=np.uint32)
        counts = np.zeros(int(self._graphs.layout.n_counters), dtype=np.uint32) if want_table else None
        self.ctx._chk(self.ctx.L.pg_batch_download_all(self.ctx.h, self.h, res.ctypes.data, ops.ctypes.data, len(ops),
                                                       counts.ctypes.data if want_table else None, sup.ctypes.data, path.ctypes.data, len(path)))
        return res[:self.n_reads], ops[:int(n_ops.value)], counts, sup[:self.n_reads], path[:int(n_path.value)]

    def align(self, flags=AF_ALL):
        self.ctx._chk(self.ctx.L.pg_batch_align(self.ctx.h, self.h, flags & 0xFFFFFFFF))

    def set_fragments(self, fragment_of_read, is_reverse_strand=None):
        fr = _u32(fragment_of_read)
        rv = None
        if is_reverse_strand is not None:
            rv = np.ascontiguousarray(is_reverse_strand, dtype=np.uint8)
        self.ctx._chk(self.ctx.L.pg_batch_set_fragments(
            self.ctx.h, self.h, _p32(fr), rv.ctypes.data_as(C.POINTER(C.c_uint8)) if rv is not None else None))

    def count(self, remove_nonuniq=True, bad_align_frac=0.8, use_support_filters=True, d_counts=None, use_kmer_filter=False):
        """Runs the count path (async); d_counts = device pointer (int) of a caller-owned uint32 table or None."""
        prm = CountParams(1 if remove_nonuniq else 0, 1 if use_support_filters else 0, bad_align_frac,
                          1 if use_kmer_filter else 0, 0)
        self.ctx._chk(self.ctx.L.pg_batch_count(self.ctx.h, self.h, C.byref(prm), d_counts))

    def download_counts(self, want_table=True, into=None):
        """-> (counts table or None, supports (SUPPORT_DTYPE), path entries).  into = (counts, supports, path) arrays to
        fill instead of fresh ones (e.g. pinned: the copies are then DMAs beside another batch's kernels)."""
        lay = self._graphs.layout
        npath = C.c_uint64()
        self.ctx._chk(self.ctx.L.pg_batch_download_counts(self.ctx.h, self.h, None, None, None, 0, C.byref(npath)))
        if into is not None:
            counts, sup, path = into
            if not want_table:
                counts = None
            if len(sup) < self.n_reads or len(path) < int(npath.value) or (counts is not None and len(counts) < int(lay.n_counters)):
                raise ValueError("download_counts: `into` arrays too small")
        else:
            counts = np.zeros(int(lay.n_counters), dtype=np.uint32) if want_table else None
            sup = np.zeros(max(self.n_reads, 1), dtype=SUPPORT_DTYPE)
            path = np.zeros(max(int(npath.value), 1), dtype=np.uint32)
        self.ctx._chk(self.ctx.L.pg_batch_download_counts(
            self.ctx.h, self.h, counts.ctypes.data if want_table else None, sup.ctypes.data, path.ctypes.data,
            len(path), C.byref(npath)))
        return counts, sup[:self.n_reads], path[:int(npath.value)]

    def download_label_sets(self, sup):
        """The reads' label sets as Python integers (bit i = label i of the read's graph): word 0 from `sup` (download_counts),
        the further words of a graph set with more than 64 labels on a graph from pg_batch_download_label_ext."""
        words = getattr(self._graphs, "label_words", 1)
        out = [int(x) for x in sup["label_mask"]]
        if words > 1 and self.n_reads:
            ext = np.zeros((self.n_reads, words - 1), dtype=np.uint64)
            self.ctx._chk(self.ctx.L.pg_batch_download_label_ext(self.ctx.h, self.h, ext.ctypes.data_as(C.POINTER(C.c_uint64)), ext.size))
            for i in range(self.n_reads):
                for w in range(words - 1):
                    out[i] |= int(ext[i, w]) << (64 * (w + 1))
        return out

    def download(self, into=None):
        """-> (results (RESULT_DTYPE), ops).  into = (results, ops) arrays to fill (e.g. pinned) instead of fresh ones."""
        cnt = C.c_uint64()
        self.ctx._chk(self.ctx.L.pg_batch_ops_count(self.ctx.h, self.h, C.byref(cnt)))
        if into is not None:
            res, ops = into
            if len(res) < self.n_reads or len(ops) < cnt.value:
                raise ValueError("download: `into` arrays too small")
        else:
            res = np.zeros(max(self.n_reads, 1), dtype=RESULT_DTYPE)
            ops = np.zeros(max(cnt.value, 1), dtype=np.uint32)
        got = C.c_uint64()
        self.ctx._chk(self.ctx.L.pg_batch_download(self.ctx.h, self.h, res.ctypes.data, ops.ctypes.data, len(ops),
                                                   C.byref(got)))
        return res[:self.n_reads], ops[:got.value]


def render_cigar(res_row, ops):
    """'<node>[<len><op>...]...' exactly as GraphAlignerImpl::extractCigar prints it."""
    out = []
    cur = None
    o0 = int(res_row["ops_off"])
    for e in range(int(res_row["n_ops"])):
        o = int(ops[o0 + e])
        node, code, ln = o >> 16, (o >> 12) & 0xF, o & 0xFFF
        if node != cur:
            if cur is not None:
                out.append("]")
            out.append("%d[" % node)
            cur = node
            last = None
        if code < 6:
            if last is not None and last[0] == code:  # pieces of one run (beyond 4 095 bases) print as one element
                last[1] += ln
            else:
                last = [code, ln]
                out.append(last)
    if cur is not None:
        out.append("]")
    return "".join(x if isinstance(x, str) else "%d%s" % (x[1], OP_CHARS[x[0]]) for x in out)


def render_cigars(res, ops, stride=128):
    """All CIGAR strings at once: (n, stride) uint8 array, slot i = NUL-padded string of read i (pg_render_cigars)."""
    L = load_library()
    res = np.ascontiguousarray(res)
    ops = np.ascontiguousarray(ops, dtype=np.uint32)
    buf = np.zeros((len(res), stride), dtype=np.uint8)
    st = L.pg_render_cigars(res.ctypes.data, len(res), ops.ctypes.data, buf.ctypes.data, stride)
    if st != PG_OK:
        raise PgError(st, "pg_render_cigars(stride=%d)" % stride)
    return buf


def results_to_dicts(res, ops):
    out = []
    for r in res:
        mm = int(r["multi_mask"])
        out.append({
            "graph_pos": int(r["graph_pos"]), "score": int(r["score"]), "mapq": int(r["mapq"]),
            "unique": bool(r["is_unique"]), "returned_reverse": bool(r["returned_reverse"]),
            "multi": [(mm >> k) & 1 for k in range(4)],
            "other_fwd_skipped": bool(mm & 0x10),  # lean stage: the forward-graph fill of the strand not returned did not run
            "strand_score": [int(r["strand_score"][0]), int(r["strand_score"][1])],
            "cigar": render_cigar(r, ops), "clipped": int(r["clipped"]), "status": int(r["status"]) & 0xFF,
            "by_path_aligner": bool(int(r["status"]) & STATUS_PATH_ALIGNER),
        })
    return out


def decode_counts(graphs, counts):
    """counts table -> per graph dict(node_counts[n,4], edge_counts {(from,to): [4]}, seq_counts {"A,B": [4]},
    tallies {aligned, mapped, bad_align, nonuniq})."""
    lay = graphs.layout
    out = []
    ek = 0
    for g in range(graphs.n):
        nb, ne = int(graphs.node_off[g]), int(graphs.node_off[g + 1])
        nodes = counts[int(lay.node_base) + 4 * nb:int(lay.node_base) + 4 * ne].reshape(-1, 4).astype(np.uint64)
        edges = {}
        for e in graphs.edges[g]:
            edges[e] = [int(x) for x in counts[int(lay.edge_base) + 4 * ek:int(lay.edge_base) + 4 * ek + 4]]
            ek += 1
        seqs = {}
        labs = graphs.labels[g]
        for m in range(int(graphs.seq_off[g + 1] - graphs.seq_off[g])):
            o = int(lay.seq_base) + 4 * (int(graphs.seq_off[g]) + m)
            c = [int(x) for x in counts[o:o + 4]]
            if c[0]:
                seqs[",".join(sorted(labs[i] for i in range(len(labs)) if (m >> i) & 1))] = c
        t = counts[int(lay.tally_base) + 4 * g:int(lay.tally_base) + 4 * g + 4]
        out.append({"node_counts": nodes, "edge_counts": edges, "seq_counts": seqs,
                    "tallies": {"aligned": int(t[0]) & 0x7FFFFFFF, "mapped": int(t[1]), "bad_align": int(t[2]),
                                "nonuniq": int(t[3]), "overflow": bool(int(t[0]) >> 31)}})
    return out


def decode_supports(graphs, graph_of_read, sup, path, label_sets=None):
    """per read: dict(status, filter, nodes {ids}, edges {(from,to)}, labels {names}).  label_sets = Batch.download_label_sets(sup)
    for graph sets with more than 64 labels on a graph (default: the 64 bits in `sup`)."""
    out = []
    for i, s in enumerate(sup):
        g = int(graph_of_read[i])
        nodes, edges = set(), set()
        prev = None
        for k in range(int(s["n_path"])):
            en = int(path[int(s["path_off"]) + k])
            nd = en & 0xFFFF
            if (en >> 30) & 1:
                nodes.add(nd)
            if k > 0 and (en >> 31) & 1:
                edges.add((prev, nd))
            prev = nd
        labs = graphs.labels[g]
        out.append({"status": int(s["status"]), "filter": int(s["filter"]), "nodes": nodes, "edges": edges,
                    "labels": {labs[b] for b in range(len(labs))
                               if ((label_sets[i] if label_sets is not None else int(s["label_mask"])) >> b) & 1}})
    return out
