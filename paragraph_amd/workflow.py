"""Python door to the many-site host workflow (libparagraph_host.so, include/paragraph_workflow.h): what
src/python/bin/multigrmpy.py gets from spawning the `grmpy` binary, as a function call.

    from paragraph_amd import workflow
    genotypes = workflow.genotype_graphs("ref.fa", "manifest.txt", ["g1.json", "g2.json"], threads=32)

There is no CPU fallback: the realignment runs on the MI355X, and a missing library is an error.
"""
import ctypes as C
import json
import os
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libparagraph_host.so")
_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise FileNotFoundError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % HOST_LIB_PATH)
        L = C.CDLL(HOST_LIB_PATH)
        L.pgw_genotype_graphs.restype = C.c_int
        L.pgw_genotype_graphs.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p,
                                          C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


def genotype_graphs_to_file(reference_fasta, manifest, graph_paths, output_path, genotyping_parameters=None, **options):
    """The call itself: BAM files in, the JSON array of genotype documents in output_path.  Nothing is read back (a caller
    that times the workflow does not want Python's JSON parser inside the timed region)."""
    L = load_library()
    paths = (C.c_char_p * len(graph_paths))(*[os.fsencode(p) for p in graph_paths])
    err = C.create_string_buffer(4096)
    rc = L.pgw_genotype_graphs(os.fsencode(reference_fasta), os.fsencode(manifest), paths, len(graph_paths),
                               os.fsencode(genotyping_parameters) if genotyping_parameters else None,
                               json.dumps(options).encode() if options else None, os.fsencode(output_path), err, len(err))
    if rc != 0:
        raise RuntimeError("pgw_genotype_graphs: " + err.value.decode(errors="replace"))


def genotype_graphs(reference_fasta, manifest, graph_paths, genotyping_parameters=None, output_path=None, **options):
    """Genotypes every graph against every sample of the manifest; returns the list of genotype documents (and leaves the
    JSON array in output_path when one is given).  Options: threads, lanes, sites_per_batch, max_reads, bad_align_frac,
    path_sequence_matching, kmer_sequence_matching, klib_sequence_matching, exact_match_shortcut, bad_align_uniq_kmer_len, packed_reads,
    devices (list of HIP ordinals the lanes are spread over; default PG_DEVICES, else device 0)."""
    keep = output_path is not None
    if not keep:
        fd, output_path = tempfile.mkstemp(suffix=".json")
        os.close(fd)
    try:
        genotype_graphs_to_file(reference_fasta, manifest, graph_paths, output_path, genotyping_parameters, **options)
        with open(output_path) as f:
            return json.load(f)
    finally:
        if not keep and os.path.exists(output_path):
            os.remove(output_path)
