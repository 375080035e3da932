// pg_general.hip -- kernels of the general form of the gssw stage (pg_general.h): one WAVEFRONT per fill (round 4; one thread
// per fill before), one thread per traceback.  Only reads the packed wavefront kernels cannot take come here (longer than PG_MAX_READ_LEN, or on a graph of
// more than 65 519 columns); it is the slow lane that keeps such a read -- and its site -- in the run.
//
// Replaces, for those reads, gssw_graph_fill + gssw_graph_trace_back (external/gssw/gssw.c:4033-4044, 3539-3560) as
// GraphAligner::alignRead drives them (src/c++/lib/grm/GraphAligner.cpp:308-404).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_general.h"

// One WAVEFRONT per fill (read x strand x graph direction).  Lane k owns the read rows [k R, (k + 1) R), R = ceil(L / 64), and
// sweeps the direction's columns in layout (= topological) order one column behind lane k - 1 -- the anti-diagonal pipeline of
// pg_fill.hip in its general form: plain 16-bit scores in 32-bit arithmetic, 32-bit column indices, the two rolling columns (H
// of the previous column / E of this one, one word per row) and the query's codes in LDS (5 L bytes: 80 KB at 16 000 bases), the lane's own (node, column) position
// instead of column meta words (a graph of this path may have more nodes than a meta word can name).  Per step a lane takes
// from the lane above what row k R - 1 of the same column left behind one step earlier: its H, the H it had in the column
// before (the diagonal input) and its F.  Node boundaries cost nothing extra in the pipeline: a lane that reaches a node's
// first column builds its rows' seeds from the predecessors' last columns -- rows it stored ITSELF when it passed them, so
// program order is all the synchronisation there is -- and at a node's last column it stores its rows' seeds and folds its
// part of the node's maximum into the node's key with an atomicMax (first cell in (column, row) order: score | inverted
// column | inverted row).  Same arithmetic, same outputs as pggen::fill (one thread per fill, which tests/test_general_cpu.py
// pins against the reference's gssw.c on the CPU and which remains the statement of record); 64 lanes instead of one.
__global__ __launch_bounds__(64) void pg_gen_fill_kernel(PgGenArgs a)
{
    extern __shared__ uint32_t gen_lds[];
    const uint32_t k = blockIdx.x >> 2, f = blockIdx.x & 3u;
    const int lane = (int)threadIdx.x;
    const PgGenRead gr = a.reads[k];
    const int dir = (int)(f >> 1), strand = (int)(f & 1u);
    PgFillSummary* out = a.fsum + (size_t)k * 4 + f;
    if ((strand && !(a.flags & PG_AF_BOTH_STRANDS)) || (dir && !(a.flags & PG_AF_REVERSE_GRAPH)))
    {
        if (lane == 0)
        {
            PgFillSummary z{};
            z.ref_end = z.end_col = -1;
            *out = z;
        }
        return;
    }
    const PgGraphDev gd = a.graphs[gr.graph];
    const uint32_t off = a.base_off[gr.read];
    const int L = (int)(a.base_off[gr.read + 1] - off);
    const char* bases = a.bases + off;
    const PgNode* nodes = a.nodes + gd.dir[dir].node_off;
    const uint32_t n_nodes = gd.dir[dir].n_nodes, ncols = gd.dir[dir].ncols;
    const size_t n_nodes_fwd = gd.dir[0].n_nodes;
    int16_t* H = dir == 0 ? (int16_t*)(a.ws + gr.h_off) + (size_t)strand * gd.dir[0].ncols * (size_t)L : nullptr;
    int16_t* seedH = (int16_t*)(a.ws + gr.seed_off) + (size_t)f * 2 * n_nodes_fwd * (size_t)L;
    int16_t* seedE = seedH + n_nodes_fwd * (size_t)L;
    unsigned long long* node_key = (unsigned long long*)(a.ws + gr.node_off + (size_t)f * n_nodes_fwd * PG_GEN_NODE_BYTES);  // [node] {key, half}
    uint32_t* colHE = gen_lds;                       // [L] H of the previous column (or the seed) | E of this column << 16
    uint8_t* qcode = (uint8_t*)(gen_lds + L);        // [L] nt code of the fill's query string (GraphAligner.cpp:315-337)
    const int R = (L + 63) / 64;
    const int j0 = lane * R, j1 = j0 + R < L ? j0 + R : L;  // this lane's rows (none for the last lanes of a short read)
    for (int j = j0; j < j1; ++j)
        qcode[j] = (uint8_t)pggen::nt_code(pggen::query_char(bases, L, dir, strand, j));  // (a lane only ever reads its own rows)
    for (uint32_t n = (uint32_t)lane; n < n_nodes; n += 64u)
    {
        node_key[2 * (size_t)n] = 0ull;
        node_key[2 * (size_t)n + 1] = 0ull;
    }
    __threadfence();
    __syncthreads();

    // the lane's position: node `id`, column `i` of it; what it hands to the lane below (from its last row, this column)
    uint32_t id = 0, i = 0;
    PgNode nd = nodes[0];
    int out_h = 0, out_diag = 0, out_f = 0;
    int nb = 0, nb_i = 0, nb_j = 0, first_half = 0;  // the lane's part of the current node's maximum
    uint64_t half = ((uint64_t)nd.len * (uint64_t)L + 1u) / 2u;
    const uint32_t steps = ncols + 63u;
    for (uint32_t t = 0; t < steps; ++t)
    {
        // (every lane takes part in the shuffles, whatever it does with them)
        const int in_h = __shfl_up(out_h, 1), in_diag = __shfl_up(out_diag, 1), in_f = __shfl_up(out_f, 1);
        const bool active = t >= (uint32_t)lane && t - (uint32_t)lane < ncols;
        if (!active)
            continue;
        if (i == 0)
        {
            // seeds: row-wise maxima over the predecessors (ascending ids) of their last column's H and next-column E
            for (int j = j0; j < j1; ++j)
            {
                int sh = 0, se = 0;
                for (uint32_t p = 0; p < nd.n_pred; ++p)
                {
                    const size_t at = (size_t)a.preds[nd.pred_off + p] * (size_t)L + (size_t)j;
                    sh = pggen::imax(sh, seedH[at]);
                    se = pggen::imax(se, seedE[at]);
                }
                colHE[j] = (uint32_t)sh | ((uint32_t)se << 16);
            }
            nb = 0;
            nb_i = 0;
            nb_j = 0;
            first_half = 0;
            half = ((uint64_t)nd.len * (uint64_t)L + 1u) / 2u;
        }
        const uint32_t rcode = pggen::nt_code(pggen::ref_char(gd, a.nodes, a.seqchars, dir, id, nd.len, i));
        int16_t* Hc = H ? H + ((size_t)nd.col_start + i) * (size_t)L : nullptr;
        // row j0 - 1 of this column belongs to the lane above (lane 0: the matrix's edge)
        int diag = lane ? in_diag : 0, hleft = lane ? in_h : 0, fv = lane ? in_f : 0;
        int hp_last = 0;
        for (int j = j0; j < j1; ++j)
        {
            const uint32_t he = colHE[j];
            const int hp = (int)(he & 0xFFFFu);  // H of the previous column (or the seed)
            const int e = (int)(he >> 16);       // E of this column
            fv = j > 0 ? pggen::imax(pggen::sat_sub(fv, PG_GAP_EXT), pggen::sat_sub(hleft, PG_GAP_OPEN)) : 0;
            const int d = j > 0 ? diag : 0;
            int h = pggen::imax(d + pggen::sub_score(rcode, qcode[j]), 0);
            h = pggen::imax(h, pggen::imax(e, fv));
            diag = hp;
            hleft = h;
            hp_last = hp;
            colHE[j] = (uint32_t)h | ((uint32_t)pggen::imax(pggen::sat_sub(e, PG_GAP_EXT), pggen::sat_sub(h, PG_GAP_OPEN)) << 16);  // H | E of the next column
            if (Hc)
                Hc[j] = (int16_t)h;
            if (h > nb)
            {  // the lane's first cell in (column, row) order holding its maximum of the node
                nb = h;
                nb_i = (int)i;
                nb_j = j;
            }
            if ((uint64_t)i * (uint64_t)L + (uint64_t)j < half && h > first_half)
                first_half = h;
        }
        if (j1 > j0)
        {
            out_h = hleft;
            out_diag = hp_last;
            out_f = fv;
        }
        else
        {  // a lane without rows passes on what it was given (nobody below it has rows either)
            out_h = in_h;
            out_diag = in_diag;
            out_f = in_f;
        }
        if (i + 1 == nd.len)
        {
            for (int j = j0; j < j1; ++j)
            {
                seedH[(size_t)id * (size_t)L + (size_t)j] = (int16_t)(colHE[j] & 0xFFFFu);
                seedE[(size_t)id * (size_t)L + (size_t)j] = (int16_t)(colHE[j] >> 16);
            }
            if (nb > 0)
                atomicMax(&node_key[2 * (size_t)id], ((unsigned long long)nb << 40) | ((unsigned long long)(0xFFFFFFu - (uint32_t)nb_i) << 16)
                                                        | (unsigned long long)(0xFFFFu - (uint32_t)nb_j));
            if (first_half > 0)
                atomicMax((unsigned int*)&node_key[2 * (size_t)id + 1], (unsigned int)first_half);
            ++id;
            i = 0;
            if (id < n_nodes)
                nd = nodes[id];
        }
        else
            ++i;
    }
    __threadfence();
    __syncthreads();

    // ---- max_node = first node with the strictly largest score (gssw.c:4015), alignsEndAtMultNodes ---------------------------
    unsigned long long best = 0;  // (score << 32) | inverted node id: the largest score, among equals the smallest node
    for (uint32_t n = (uint32_t)lane; n < n_nodes; n += 64u)
    {
        const unsigned long long key = __hip_atomic_load(&node_key[2 * (size_t)n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long sc = key >> 40;
        if (sc)
        {
            const unsigned long long v = (sc << 32) | (unsigned long long)(0xFFFFFFFFu - n);
            best = v > best ? v : best;
        }
    }
    for (int o = 32; o >= 1; o >>= 1)
    {
        const unsigned long long v = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), o) << 32)
            | (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)best, o);
        best = v > best ? v : best;
    }
    const int best_score = (int)(best >> 32);
    // A fill whose top score reaches 251 was redone by gssw in 16-bit words (gssw.c:380, 4100-4104), and the scan of
    // alignsEndAtMultNodes then reads len * readLen BYTES of that matrix (GraphAligner.cpp:170-212): only the low bytes of the
    // first half of a node's cells can hold a top score, which must be <= 255.
    // (an all-zero fill -- e.g. the reverse complement of a lower-case read, all N -- "ends" in every node: the scalar form and
    // the reference count it like any other top score)
    uint32_t hits = 0;
    for (uint32_t n = (uint32_t)lane; n < n_nodes; n += 64u)
        {
            const unsigned long long key = __hip_atomic_load(&node_key[2 * (size_t)n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t fh = (uint32_t)__hip_atomic_load(&node_key[2 * (size_t)n + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (best_score < 251)
                hits += (int)(key >> 40) == best_score;
            else if (best_score <= 255)
                hits += (int)fh == best_score;
        }
    for (int o = 32; o >= 1; o >>= 1)
        hits += (uint32_t)__shfl_xor((int)hits, o);
    if (lane == 0)
    {
        PgFillSummary fs{};
        fs.score = best_score;
        fs.ref_end = -1;
        fs.end_col = -1;
        if (best_score > 0)
        {
            const uint32_t bn = 0xFFFFFFFFu - (uint32_t)best;
            const unsigned long long key = __hip_atomic_load(&node_key[2 * (size_t)bn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t bi = 0xFFFFFFu - (uint32_t)((key >> 16) & 0xFFFFFFu), bj = 0xFFFFu - (uint32_t)(key & 0xFFFFu);
            fs.max_node = (int32_t)bn;
            fs.ref_end = (int32_t)bi;
            fs.read_end = (int32_t)bj;
            fs.end_col = (int32_t)(nodes[bn].col_start + bi);
        }
        fs.multi = hits > 1 ? 1 : 0;
        *out = fs;
    }
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
    // one wavefront per fill; its two rolling columns and query codes live in LDS (5 bytes per base of the launch's longest read)
    const size_t lds = (size_t)4 * args.max_len + (((size_t)args.max_len + 3) & ~(size_t)3);  // (H | E) words + query codes
    if (lds > 48 * 1024)
    {
        hipError_t ea = hipFuncSetAttribute((const void*)pg_gen_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess)
            return ea;
    }
    hipLaunchKernelGGL(pg_gen_fill_kernel, dim3(args.n * 4), dim3(64), lds, stream, args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(pg_gen_trace_kernel, dim3((args.n + 63) / 64), dim3(64), 0, stream, args);
    return hipGetLastError();
}
