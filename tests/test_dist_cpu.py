"""world_size-2 gloo test of the N>1 path's host logic: sites are partitioned across ranks, every rank fills
the counter table for its shard (here with the CPU checker standing in for the device kernels -- the table
layout and the reduce are what is under test), the tables are all-reduced, and the result must equal the
single-process table."""
import os
import random
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_sites(seed, n_sites):
    from oracle.oracle import PortOracle
    from oracle import counts as oc
    from tests import fuzzgen
    rng = random.Random(seed)
    chk = PortOracle()
    sites = []
    for _ in range(n_sites):
        seqs, edges = fuzzgen.rand_graph(rng, max_len=30, max_nodes=5)
        labels, names = fuzzgen.rand_labels(rng, edges)
        reads = [fuzzgen.rand_read(rng, seqs, edges, min_len=10, max_len=60) for _ in range(rng.randint(4, 10))]
        frag = fuzzgen.rand_fragments(rng, len(reads))
        al = chk.align_batch(seqs, edges, reads)
        recs = [{"pos": a["graph_pos"], "cigar": a["cigar"], "aligned": a["score"] > 0, "unique": a["unique"],
                 "graph_reverse": a["returned_reverse"], "read_len": len(r), "fragment": frag[i]}
                for i, (a, r) in enumerate(zip(al, reads))]
        sites.append((oc.CountGraph(seqs, edges, labels, names), recs, sum(len(r) for r in reads) * sum(map(len, seqs))))
    return sites


def _table_for(sites, which):
    """Flat [nodes*4][edges*4] table over ALL sites with only the sites in `which` filled."""
    from oracle import counts as oc
    node_parts, edge_parts = [], []
    for i, (g, recs, _) in enumerate(sites):
        if i in which:
            out = oc.port_count_site(g, recs)
            node_parts.append(np.asarray(out["node_counts"], dtype=np.int64).reshape(-1))
            edge_parts.append(np.asarray(out["edge_counts"], dtype=np.int64).reshape(-1))
        else:
            node_parts.append(np.zeros(4 * len(g.nodes), dtype=np.int64))
            edge_parts.append(np.zeros(4 * len(g.edges), dtype=np.int64))
    return np.concatenate(node_parts + edge_parts)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from paragraph_amd import dist as pd
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    sites = _make_sites(99, 12)
    parts = pd.partition_sites([w for _, _, w in sites], world)
    mine = set(int(i) for i in parts[rank])
    t = torch.from_numpy(_table_for(sites, mine))

    class Drained:  # stands in for the device context: the host backend drains the compute streams before it reduces
        calls = 0

        def sync_compute(self):
            Drained.calls += 1

    red = pd.CountReduce(Drained(), None)
    assert red.active and red.blocking  # gloo: no stream to order against, the blocking form
    red.acquire(t)
    red.reduce(t)
    red.wait()
    assert Drained.calls == 1 and red.reduces == 1
    q.put((rank, t.numpy().copy(), sorted(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    from paragraph_amd import dist as pd
    rng = np.random.RandomState(3)
    w = rng.randint(1, 1000, size=200)
    for world in (1, 2, 4, 8):
        parts = pd.partition_sites(w, world)
        allidx = np.sort(np.concatenate(parts))
        assert (allidx == np.arange(200)).all()
        loads = np.array([w[p].sum() for p in parts], dtype=np.float64)
        assert loads.max() <= loads.mean() * 1.05 + w.max()
    fr = np.array([0, 0, 1, 2, 2, 2, 3, 4, 4])
    parts = pd.partition_fragments(fr, 2)
    assert sorted(np.concatenate(parts).tolist()) == list(range(9))
    for p in parts:
        for q in parts:
            if p is not q:
                assert not set(fr[p]) & set(fr[q])  # mates never split


def test_two_rank_count_reduce_gloo():
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    sites = _make_sites(99, 12)
    want = _table_for(sites, set(range(len(sites))))
    shards = sorted(got)
    assert sorted(shards[0][2] + shards[1][2]) == list(range(12)) and shards[0][2] and shards[1][2]
    for _, table, _ in shards:
        assert (table == want).all()
    assert want.sum() > 0
