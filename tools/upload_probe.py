"""Times the host <-> device legs around the resident-data bench (first call vs steady state)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from paragraph_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    ws = float(sys.argv[2]) if len(sys.argv) > 2 else 64
    ctx = capi.Context(0, workspace_bytes=int(ws * (1 << 30)))
    site, arr = synth.config2_reads_packed(n, read_len=150, seed=2)
    graphs = ctx.upload_graphs([(site.seqs, site.edges)])
    graphs.set_labels([site.labels])
    t = time.perf_counter()
    packed = synth.packed_to_capi(arr)
    print("packed_to_capi %.3fs" % (time.perf_counter() - t))
    frag = np.arange(n, dtype=np.uint32) // 2
    for rep in range(3):
        b = ctx.new_batch()
        t = time.perf_counter()
        b.upload(graphs, packed)
        ctx.sync()
        t1 = time.perf_counter()
        b.set_fragments(frag)
        ctx.sync()
        t2 = time.perf_counter()
        b.align(capi.AF_ALL)
        b.count()
        ctx.sync()
        t3 = time.perf_counter()
        res, ops = b.download()
        t4 = time.perf_counter()
        b.upload(graphs, packed)
        ctx.sync()
        t5 = time.perf_counter()
        print("rep %d: upload %.3fs set_fragments %.3fs align+count %.3fs download %.3fs re-upload(same batch) %.3fs" %
              (rep, t1 - t, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
        b.close()


if __name__ == "__main__":
    main()
