"""BASELINE configs[3] in end-to-end form: the site list of an e2e data set sharded over the ranks of one node (rank r takes
sites r, r + W, ...; each rank drives its own GPU through the workflow), then ONE all-reduce of the per-site edge-count table.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/e2e/run_ranks.py <data dir made by make_sites.py> [threads_per_rank]

With one GPU per rank the reduce runs over RCCL ("nccl"); when ranks have to share a GPU (a 1-GPU box) it falls back to
gloo so the path can still be exercised.  Rank 0 prints one JSON line (wall clock = slowest rank, sites/s over all ranks)
and checks the reduced table against the genotype documents' own counts."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    data = sys.argv[1]
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    n_gpus = torch.cuda.device_count()
    shared_gpu = n_gpus < world
    device = local % max(n_gpus, 1)
    os.environ["PG_DEVICE"] = str(device)  # which GPU libparagraph_host's device context opens
    from paragraph_amd import workflow
    if world > 1:
        dist.init_process_group("gloo" if shared_gpu else "nccl", rank=rank, world_size=world)
    graphs = [l.strip() for l in open(os.path.join(data, "graphs.txt")) if l.strip()]
    # the table layout is the same on every rank: one slot per edge of every graph, in file order
    offsets, total = [], 0
    edge_keys = []
    for g in graphs:
        edges = ["%s_%s" % (e["from"], e["to"]) for e in json.load(open(g))["edges"]]
        offsets.append(total)
        edge_keys.append(edges)
        total += len(edges)
    mine = list(range(rank, len(graphs), world))
    table = torch.zeros(total, dtype=torch.int32)
    if world > 1:
        dist.barrier()
    t0 = time.time()
    docs = workflow.genotype_graphs(os.path.join(data, "ref.fa"), os.path.join(data, "manifest.txt"), [graphs[i] for i in mine],
                                    threads=threads, lanes=max(1, min(8, threads // 4)), sites_per_batch=512)
    for i, doc in zip(mine, docs):
        sample = next(iter(doc["samples"].values()))
        counts = {}
        for bp in sample["breakpoints"].values():
            counts.update(bp["counts"]["edges"])
        for k, key in enumerate(edge_keys[i]):
            table[offsets[i] + k] = counts.get(key, 0)
    if world > 1:
        if not shared_gpu:
            table = table.cuda(device)
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
        table = table.cpu()
    elapsed = torch.tensor([time.time() - t0], dtype=torch.float64)
    if world > 1:
        if not shared_gpu:
            elapsed = elapsed.cuda(device)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        elapsed = elapsed.cpu()
    if rank == 0:
        truth = {t["ID"]: t["gt"] for t in json.load(open(os.path.join(data, "truth.json")))}
        ok = sum(1 for d in docs if d["samples"]["SYN"]["gt"]["GT"] == truth[d["graphinfo"]["ID"]])
        print(json.dumps({"ranks": world, "backend": "none" if world == 1 else ("gloo" if shared_gpu else "nccl"), "sites": len(graphs),
                          "seconds": round(float(elapsed[0]), 3), "sites_per_s": round(len(graphs) / float(elapsed[0])),
                          "edge_table_entries": total, "edge_table_sum": int(table.sum()), "rank0_sites": len(mine),
                          "rank0_concordant": ok}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
