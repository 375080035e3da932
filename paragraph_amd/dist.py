"""Multi-GPU plumbing: one process per GPU, sites/fragments sharded across ranks, and the path's only
collective -- the integer all-reduce of the per-site counter table (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests).

Sharding rule: every (site, read) alignment is independent; counts are per *fragment*
(src/c++/lib/paragraph/ReadCounting.cpp:52-94, src/c++/lib/common/Fragment.cpp:141-181), so whole sites --
or, inside a hot site, whole fragments -- go to one rank, never single reads."""
import os

import numpy as np


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def partition_sites(weights, world):
    """Deterministic balanced partition of sites by weight (e.g. sum of read_len * graph_len per site):
    longest-processing-time greedy.  Returns a list of `world` sorted index arrays covering every site once."""
    w = np.asarray(weights, dtype=np.float64)
    order = np.lexsort((np.arange(len(w)), -w))  # heaviest first, index as tie-break
    loads = np.zeros(world)
    parts = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(loads))  # first minimum: deterministic
        parts[r].append(int(i))
        loads[r] += w[i]
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def partition_fragments(fragment_of_read, world):
    """Splits the reads of ONE hot site by fragment id so that mates stay together: rank = fragment id mod world.
    Returns a list of read-index arrays."""
    f = np.asarray(fragment_of_read, dtype=np.int64)
    return [np.nonzero(f % world == r)[0] for r in range(world)]


def allreduce_counts(table):
    """In-place SUM all-reduce of a counter table (torch tensor, int32/int64) over the default process group -- the only
    collective of the path.  With the "nccl" backend (= RCCL) the tensor is reduced where it lives, in HBM over xGMI; a
    device tensor under a host backend ("gloo": CPU tests, or several ranks sharing one GPU) takes one hop through host
    memory.  The caller orders it against the kernels that filled the table (pg_ctx_sync_compute) -- the library's streams
    are not torch's."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return table
    if table.is_cuda and dist.get_backend() != "nccl":
        host = table.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        table.copy_(host)
    else:
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
    return table
