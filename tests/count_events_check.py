"""Run by tests/test_gpu_scale.py::test_count_stream_events_order_a_foreign_stream in a process of its own (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch  # before the library: two HIP runtimes in one process, torch's first (bench.py's order)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from paragraph_amd import capi, synth
    ctx = capi.Context(0)
    site, arr = synth.config2_reads_packed(50000, read_len=150, seed=5)
    G = ctx.upload_graphs([(site.seqs, site.edges)])
    G.set_labels([site.labels])
    b = ctx.new_batch()
    b.upload(G, synth.packed_to_capi(arr))
    b.set_fragments(np.arange(len(arr), dtype=np.uint32) // 2)
    n = int(G.layout.n_counters)
    table = torch.zeros(n, dtype=torch.int32, device=dev)
    snap = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    ev_counted, ev_read = torch.cuda.Event(), torch.cuda.Event()
    ev_counted.record()
    ev_read.record()  # torch creates the handles at the first record
    for k in range(3):
        ctx.counts_zero(table.data_ptr(), n)
        b.align(capi.AF_ALL)
        b.count(remove_nonuniq=True, d_counts=table.data_ptr())
        ctx.count_record(ev_counted.cuda_event)
        side.wait_event(ev_counted)
        with torch.cuda.stream(side):
            snap[k].copy_(table)
            ev_read.record(side)
        ctx.count_wait(ev_read.cuda_event)  # the next counts_zero must not run under the copy
    ctx.sync()
    torch.cuda.synchronize()
    ref = table.cpu().numpy()
    assert ref.sum() > 0
    for k in range(3):
        assert np.array_equal(snap[k].cpu().numpy(), ref), k
    assert len({ctx.native_stream(w) for w in (0, 1, 2)}) == 3
    b.close()
    G.close()
    ctx.close()
    print("count events ok")


if __name__ == "__main__":
    main()
