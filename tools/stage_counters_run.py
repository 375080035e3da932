"""One pass of every kernel north_star names besides the fill / traceback -- the seed stages (path, k-mer), the klib stage's five
kernels, the count path (support + fragment kernels) and the cascade's hand-over kernels -- on config-2 reads, for a rocprofv3
--pmc / --kernel-trace collection (tools/stage_counters.sh).  Usage: python tools/stage_counters_run.py [n_reads]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paragraph_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    ctx = capi.Context(0, workspace_bytes=32 << 30)
    site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
    graphs = ctx.upload_graphs([(site.seqs, site.edges)])
    graphs.set_labels([site.labels])
    paths = [[[0, 1, 2], [0, 2]]]
    graphs.build_path_index(32)
    graphs.build_kmer_index(paths, 16)
    graphs.build_klib_index(paths)
    b = ctx.new_batch()
    b.upload(graphs, synth.packed_to_capi(arr))
    b.set_fragments(np.arange(n, dtype=np.uint32) // 2)
    keep = capi.AF_CIGAR | capi.AF_BOTH_STRANDS | capi.AF_REVERSE_GRAPH | capi.AF_KEEP_RESULTS
    for rep in range(2):  # the first pass warms code objects and caches; the collection keeps the last launch of every kernel
        b.set_active(None)
        b.path_align(fetch_flags=False)
        b.kmer_align(fetch_flags=False)
        b.klib_align(fetch_flags=False)
        # the composed cascade with the hand-over on the device (retire + group list + item build kernels)
        b.set_active(None)
        b.path_align(fetch_flags=False)
        b.count(remove_nonuniq=True, bad_align_frac=0.8)
        b.retire_mapped()
        b.align(keep)
        b.count(remove_nonuniq=True, bad_align_frac=0.8)
        ctx.sync()
    res, ops, _, sup, _ = b.download_all(want_table=False)
    print("reads %d mapped %d by_path %d" % (n, int((sup["status"] == 1).sum()), int(((res["status"] & capi.STATUS_PATH_ALIGNER) != 0).sum())))


if __name__ == "__main__":
    main()
