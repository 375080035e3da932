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
    memory.  The caller orders it against the kernels that filled the table: CountReduce below (stream-ordered, nothing
    blocks the host) or pg_ctx_sync_compute (blocking) -- the library's streams are not torch's.  A world of ONE rank under
    "nccl" still issues the collective (the same code path as eight ranks); under a host backend it is skipped."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return table
    if dist.get_world_size() == 1 and not (table.is_cuda and dist.get_backend() == "nccl"):
        return table
    if table.is_cuda and dist.get_backend() != "nccl":
        host = table.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        table.copy_(host)
    else:
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
    return table


class CountReduce:
    """The all-reduce of the counter table, ordered against the library's count stream by events only -- the host never
    waits inside a step.  Per step (table t of a small ring of tables taking turns):

        red.acquire(t)                  count stream waits for the event behind t's previous reduce (table free again)
        ctx.counts_zero(t) ; batch.align() ; batch.count(d_counts=t)
        red.reduce(t)                   event on the count stream -> the reduce stream waits for it -> all_reduce(t) there
                                        (torch's NCCL = RCCL stream takes over from the reduce stream and hands back to it)
                                        -> event behind the reduce, kept for the next acquire(t)

    With two tables the fills, traceback and count of step n + 1 run while reduce n is on the links; nothing but
    red.wait() / a device synchronisation blocks the host.  Under a host backend ("gloo": ranks sharing one GPU in the
    tests) the reduce needs the table on the host, so that form drains the compute streams first (blocking=True)."""

    def __init__(self, ctx, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.ctx = torch, dist, ctx
        self.active = dist.is_available() and dist.is_initialized()
        self.blocking = self.active and dist.get_backend() != "nccl"
        self.done = {}
        self.reduces = 0
        if self.active and not self.blocking:
            self.count_stream = torch.cuda.ExternalStream(ctx.native_stream(1), device=device)  # the library's stream, not torch's
            self.reduce_stream = torch.cuda.Stream(device=device)

    def acquire(self, table):
        if self.active and not self.blocking:
            ev = self.done.pop(table.data_ptr(), None)
            if ev is not None:
                self.count_stream.wait_event(ev)

    def reduce(self, table):
        if not self.active:
            return
        self.reduces += 1
        if self.blocking:
            self.ctx.sync_compute()
            allreduce_counts(table)
            if table.is_cuda:
                self.torch.cuda.synchronize()
            return
        ev = self.count_stream.record_event()
        self.reduce_stream.wait_event(ev)
        with self.torch.cuda.stream(self.reduce_stream):
            self.dist.all_reduce(table, op=self.dist.ReduceOp.SUM)
            self.done[table.data_ptr()] = self.reduce_stream.record_event()

    def wait(self, table=None):
        """Blocks the host until the reduce of `table` (or every outstanding one) is complete."""
        if not self.active or self.blocking:
            return
        for ptr, ev in list(self.done.items()):
            if table is None or ptr == table.data_ptr():
                ev.synchronize()
