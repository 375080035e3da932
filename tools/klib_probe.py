"""Times the klib stage alone on config-2 reads (two paths).  Usage: python tools/klib_probe.py [n_reads] [read_len]"""
import json
import sys
import time

sys.path.insert(0, ".")
from paragraph_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    ctx = capi.Context(0, workspace_bytes=8 << 30)
    site, arr = synth.config2_reads_packed(n, read_len=L, seed=2)
    graphs = ctx.upload_graphs([(site.seqs, site.edges)])
    graphs.build_klib_index([[[0, 1, 2], [0, 2]]])
    b = ctx.new_batch()
    b.upload(graphs, synth.packed_to_capi(arr))
    b.klib_align()
    ctx.sync()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        b.klib_align()
        ctx.sync()
        best = min(best, time.perf_counter() - t)
    print(json.dumps({"reads": n, "read_len": L, "klib_s": best, "reads_per_s": n / best, "packed": graphs.klib_used_packed_kernels()}))


if __name__ == "__main__":
    main()
