"""gssw stage throughput by read length on the config-2 graph (byte variants up to 250 bp, the 16-bit "wide" variants
251-512 bp): reads/s and DP cell updates/s.  Usage: python tools/readlen_probe.py [n_reads] [len,len,...]  (prints one JSON object; PG_WIDE16=1 = the 16-lane wide kernels)"""
import json
import sys
import time

import os  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paragraph_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    lens = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100, 150, 200, 250, 251, 300, 400, 480]
    ctx = capi.Context(0, workspace_bytes=64 << 30)
    out = {"reads": n, "rows": []}
    for read_len in lens:
        site, arr = synth.config2_reads_packed(n, read_len=read_len, seed=2)
        graphs = ctx.upload_graphs([(site.seqs, site.edges)])
        b = ctx.new_batch()
        b.upload(graphs, synth.packed_to_capi(arr))
        b.align(capi.AF_ALL)
        ctx.sync()
        ctx.timing_enable(True)
        ctx.timing_reset()
        t = time.perf_counter()
        for _ in range(3):
            b.align(capi.AF_ALL)
        ctx.sync()
        s = (time.perf_counter() - t) / 3
        tm = ctx.timing()
        ctx.timing_enable(False)
        cells = 4.0 * n * read_len * site.total_len
        out["rows"].append({"read_len": read_len, "graph_len": int(site.total_len), "s_per_batch": round(s, 5), "reads_per_s": round(n / s),
                            "tcups": round(cells / s / 1e12, 3),
                            "device": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()}})
        b.close()
        graphs.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
