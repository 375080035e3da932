// pg_general.hip -- kernels of the general form of the gssw stage (pg_general.h): one thread per fill, one thread per
// traceback.  Only reads the packed wavefront kernels cannot take come here (longer than PG_MAX_READ_LEN, or on a graph of
// more than 65 519 columns); it is the slow lane that keeps such a read -- and its site -- in the run.
//
// Replaces, for those reads, gssw_graph_fill + gssw_graph_trace_back (external/gssw/gssw.c:4033-4044, 3539-3560) as
// GraphAligner::alignRead drives them (src/c++/lib/grm/GraphAligner.cpp:308-404).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_general.h"

__global__ __launch_bounds__(64) void pg_gen_fill_kernel(PgGenArgs a)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = tid >> 2, f = tid & 3u;
    if (k >= a.n)
        return;
    const PgGenRead gr = a.reads[k];
    const int dir = (int)(f >> 1), strand = (int)(f & 1u);
    PgFillSummary* out = a.fsum + (size_t)k * 4 + f;
    if ((strand && !(a.flags & PG_AF_BOTH_STRANDS)) || (dir && !(a.flags & PG_AF_REVERSE_GRAPH)))
    {
        PgFillSummary z{};
        z.ref_end = z.end_col = -1;
        *out = z;
        return;
    }
    const PgGraphDev gd = a.graphs[gr.graph];
    const uint32_t off = a.base_off[gr.read];
    const int L = (int)(a.base_off[gr.read + 1] - off);
    const size_t n_nodes = gd.dir[0].n_nodes;
    int16_t* H = dir == 0 ? (int16_t*)(a.ws + gr.h_off) + (size_t)strand * gd.dir[0].ncols * (size_t)L : nullptr;
    int16_t* seedH = (int16_t*)(a.ws + gr.seed_off) + (size_t)f * 2 * n_nodes * (size_t)L;
    int16_t* seedE = seedH + n_nodes * (size_t)L;
    int16_t* colH = (int16_t*)(a.ws + gr.col_off) + (size_t)f * 2 * (size_t)L;
    int32_t* node_max = (int32_t*)(a.ws + gr.node_off) + (size_t)f * 2 * n_nodes;
    pggen::fill(gd, a.nodes, a.preds, a.seqchars, dir, strand, a.bases + off, L, H, seedH, seedE, colH, colH + L, node_max, out);
}

__global__ __launch_bounds__(64) void pg_gen_trace_kernel(PgGenArgs a)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n)
        return;
    const PgGenRead gr = a.reads[k];
    const PgGraphDev gd = a.graphs[gr.graph];
    const uint32_t off = a.base_off[gr.read];
    const int L = (int)(a.base_off[gr.read + 1] - off);
    uint32_t* scratch = (uint32_t*)(a.ws + gr.ops_off);
    pg_result res;
    const uint32_t n = pggen::pick_and_trace(gd, a.nodes, a.preds, a.seqchars, a.bases + off, L, a.flags, a.fsum + (size_t)k * 4,
                                             (const int16_t*)(a.ws + gr.h_off), (const int16_t*)(a.ws + gr.seed_off), scratch, &res);
    const unsigned long long base = atomicAdd(a.ops_counter, (unsigned long long)n);
    const uint32_t cap = pg_gen_ops_cap((uint32_t)L);
    for (uint32_t e = 0; e < n; ++e)
        a.ops[base + e] = scratch[cap - n + e];
    res.ops_off = (uint32_t)base;
    a.results[gr.read] = res;
}

hipError_t pg_launch_general(const PgGenArgs& args, hipStream_t stream)
{
    if (args.n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(pg_gen_fill_kernel, dim3((args.n * 4 + 63) / 64), dim3(64), 0, stream, args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(pg_gen_trace_kernel, dim3((args.n + 63) / 64), dim3(64), 0, stream, args);
    return hipGetLastError();
}
