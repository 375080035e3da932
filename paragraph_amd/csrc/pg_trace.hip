// pg_trace.hip -- strand pick + traceback kernel: one thread per read.
//
// Replaces
//   GraphAligner::alignRead strand/uniqueness logic     src/c++/lib/grm/GraphAligner.cpp:340-401
//   gssw_graph_trace_back_internal                       external/gssw/gssw.c:2621-3537
//   gssw_alignment_trace_back_byte                       external/gssw/gssw.c:1112-1818
//   gssw_cigar_push_back/front run-length encoding       external/gssw/gssw.c:3679-3701
//
// Only ONE traceback per read is needed: the strand choice depends on the four fills' scores and
// multi flags only (GraphAligner.cpp:340-356), never on the mappings themselves.
//
// The fill kernel stores H only (one byte per cell).  gssw's traceback reads its E and F matrices as
// well; the decisions it takes from them are re-derived here from H:
//   * "score == F(i,j)"  <=>  exists k>=1 with H(i,j-k) - go - (k-1)*ge == score  (F <= H always, so a hit
//     means equality with the maximum); k is bounded by (j - score - go + ge) / (1 + ge) because
//     H(i,j-k) <= j-k+1.
//   * inside a gap gssw tests "gap open" (score == H(prev) - go) before "gap extend"; when the open
//     test fails the extend test is the only consistent continuation, so it is taken without
//     reading E/F.
//   * "score == E(i,j)" is what remains when the diagonal and the F test fail for i > 0; on a node's
//     first column E(0,j) is the seed max_p Enext_p(j), read from the seed region the fill kernel
//     wrote, and the cross-node open/extend choice uses H_p / Enext_p of each predecessor in
//     ascending id order exactly as gssw.c:2966-3161 does.
// The randomized parity tests (tests/test_gpu_parity.py) pin this against the reference's own gssw.c.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_device.h"
#include "pg_kernels.h"

namespace
{
__device__ __forceinline__ uint32_t nt_code(uint32_t c)
{
    switch (c)
    {
    case 'A':
    case 'a':
    case 'U':
    case 'u':
        return 0;
    case 'C':
    case 'c':
        return 1;
    case 'G':
    case 'g':
        return 2;
    case 'T':
    case 't':
        return 3;
    default:
        return 4;
    }
}
__device__ __forceinline__ uint32_t upper_c(uint32_t c) { return (c >= 'a' && c <= 'z') ? c - 32u : c; }
__device__ __forceinline__ uint32_t comp_c(uint32_t c)
{
    switch (c)
    {
    case 'A':
        return 'T';
    case 'C':
        return 'G';
    case 'G':
        return 'C';
    case 'T':
        return 'A';
    default:
        return 'N';
    }
}
__device__ __forceinline__ int sub_score(uint32_t a, uint32_t b)
{
    return (a == 4u || b == 4u) ? 0 : (a == b ? 1 : -4);
}

struct Emitter
{
    pg_op* slot;  // written from the tail backwards
    uint32_t cap;
    uint32_t n;
    uint32_t last_node, last_op, last_len;
    bool overflow;
    bool node_has_ops;  // an element was emitted for the node currently being traced

    __device__ void flush()
    {
        if (last_op != 0xFFu)
        {
            if (n < cap)
            {
                slot[cap - 1 - n] = (last_node << 20) | (last_op << 16) | (last_len & 0xFFFFu);
                ++n;
            }
            else
                overflow = true;
        }
    }
    __device__ void emit(uint32_t node, uint32_t op, uint32_t len)
    {
        node_has_ops = true;
        if (last_op == op && last_node == node)
        {
            last_len += len;
            return;
        }
        flush();
        last_node = node;
        last_op = op;
        last_len = len;
    }
};
}  // namespace

template <int C, bool WIDE>
__global__ __launch_bounds__(64) void pg_trace_kernel(PgTraceArgs a)
{
    const uint32_t tid = blockIdx.x * 64u + threadIdx.x;
    if (tid >= a.n_pairs * PG_GROUPS)
        return;
    const uint32_t pair = a.pair_begin + tid / PG_GROUPS;
    const uint32_t grp = tid % PG_GROUPS;
    const PgWorkItem* fw = a.items + 2 * (size_t)pair;
    const uint32_t ridx = fw->read[grp];
    if (ridx == PG_NONE)
        return;
    const uint32_t off = a.base_off[ridx];
    const int L = (int)(a.base_off[ridx + 1] - off);
    const char* __restrict__ bases = a.bases + off;

    const PgFillSummary* fsF = a.fillsum + ((size_t)(2 * pair) * PG_GROUPS + grp) * 2;
    const PgFillSummary* fsR = a.fillsum + ((size_t)(2 * pair + 1) * PG_GROUPS + grp) * 2;
    const bool both = (a.flags & PG_AF_BOTH_STRANDS) != 0;
    const bool revg = (a.flags & PG_AF_REVERSE_GRAPH) != 0;

    // ---- GraphAligner.cpp:340-356 ------------------------------------------------------------------
    const int m0 = fsF[0].multi;
    const int m1 = both ? fsF[1].multi : 0;
    const int m2 = revg ? fsR[0].multi : 0;
    const int m3 = (revg && both) ? fsR[1].multi : 0;
    const bool fwd_unique = !m0 && !m2;
    const bool rev_unique = !m1 && !m3;
    bool return_reverse = false;
    if (!fwd_unique && rev_unique && both)
        return_reverse = true;
    else if (fwd_unique && !rev_unique)
        return_reverse = false;
    else if (both)
        return_reverse = fsF[0].score < fsF[1].score;
    const int s = return_reverse ? 1 : 0;
    const bool unique = return_reverse ? rev_unique : fwd_unique;
    const PgFillSummary fs = fsF[s];

    pg_result res;
    res.graph_pos = 0;
    res.score = (int16_t)fs.score;
    res.mapq = unique ? 60 : 0;
    res.is_unique = unique ? 1 : 0;
    res.returned_reverse = return_reverse ? 1 : 0;
    res.multi_mask = (uint8_t)(m0 | (m1 << 1) | (m2 << 2) | (m3 << 3));
    res.n_ops = 0;
    res.ops_off = 0;
    res.strand_score[0] = (int16_t)fsF[0].score;
    res.strand_score[1] = (int16_t)(both ? fsF[1].score : -1);
    res.clipped = 0;
    res.status = 0;

    if (fs.score <= 0)
    {
        // all-zero fill: gssw returns an empty CIGAR at position 0 (gssw.c:2728-2732, 2778)
        res.status = 1;
        a.results[ridx] = res;
        return;
    }

    const PgGraphDev gdev = a.graphs[fw->graph];
    const PgNode* __restrict__ nodes = a.nodes + gdev.dir[0].node_off;
    const char* __restrict__ refc = a.seqchars + gdev.seq_off;
    const uint8_t* __restrict__ trace = a.workspace + fw->trace_off;
    const uint32_t* __restrict__ seed = (const uint32_t*)(a.workspace + fw->seed_off);

    auto qchar = [&](int j) -> uint32_t {
        return s == 0 ? upper_c((uint8_t)bases[j]) : comp_c((uint8_t)bases[L - 1 - j]);
    };
    // byte variants: trace [step][C/2][lane] dwords of bytes (A_r, A_r+1, B_r, B_r+1), seed [node][lane][C] dwords of
    // bytes (H_A, H_B, Enext_A, Enext_B); wide variants: trace [step][C][lane] dwords (A_r | B_r << 16), seed
    // [node][lane][2C] dwords (H_A | H_B << 16), (Enext_A | Enext_B << 16)
    auto Hcell = [&](uint32_t col, int j) -> int {
        const uint32_t kq = (uint32_t)j / C, r = (uint32_t)j % C;
        if (WIDE)
        {
            const size_t dw = ((size_t)(col + kq) * C + r) * 64 + (grp * 16 + kq);
            return (int)((const uint16_t*)trace)[dw * 2 + (uint32_t)s];
        }
        const size_t dw = ((size_t)(col + kq) * (C / 2) + r / 2) * 64 + (grp * 16 + kq);
        return (int)trace[dw * 4 + (r & 1u) + 2u * (uint32_t)s];
    };
    auto seedH = [&](uint32_t node, int j) -> int {
        const uint32_t kq = (uint32_t)j / C, r = (uint32_t)j % C;
        if (WIDE)
            return (int)((seed[((size_t)node * 64 + (grp * 16 + kq)) * (2 * C) + 2 * r] >> (16 * s)) & 0xFFFFu);
        return (int)((seed[((size_t)node * 64 + (grp * 16 + kq)) * C + r] >> (8 * s)) & 0xFFu);
    };
    auto seedE = [&](uint32_t node, int j) -> int {
        const uint32_t kq = (uint32_t)j / C, r = (uint32_t)j % C;
        if (WIDE)
            return (int)((seed[((size_t)node * 64 + (grp * 16 + kq)) * (2 * C) + 2 * r + 1] >> (16 * s)) & 0xFFFFu);
        return (int)((seed[((size_t)node * 64 + (grp * 16 + kq)) * C + r] >> (16 + 8 * s)) & 0xFFu);
    };

    Emitter em;
    em.cap = pg_ops_cap(WIDE ? PG_VAR_WIDE + C : C);
    em.slot = a.ops_scratch + (size_t)tid * em.cap;
    em.n = 0;
    em.last_op = 0xFFu;
    em.last_node = 0;
    em.last_len = 0;
    em.overflow = false;
    em.node_has_ops = false;

    uint32_t n = (uint32_t)fs.max_node;
    int i = fs.ref_end;
    int j = fs.read_end;
    int sc = fs.score;
    bool inE = false, inF = false;
    int status = 0;
    uint32_t clipped = 0;
    if (L - 1 - j > 0)
    {
        em.emit(n, PG_OPC_S, (uint32_t)(L - 1 - j));
        clipped += (uint32_t)(L - 1 - j);
    }

    for (int guard = 0; guard < 0x7fffffff; ++guard)
    {
        const PgNode nd = nodes[n];
        const uint32_t c0 = nd.col_start;
        // ---- within-node traceback (gssw.c:1214-1808) ---------------------------------------------
        while (sc > 0 && i >= 0 && j >= 0)
        {
            if (inE)
            {
                if (i == 0)
                    break;
                const int hup = Hcell(c0 + i - 1, j);
                em.emit(n, PG_OPC_D, 1);
                --i;
                if (sc == hup - PG_GAP_OPEN)
                {
                    sc += PG_GAP_OPEN;
                    inE = false;
                }
                else
                    sc += PG_GAP_EXT;
                continue;
            }
            if (inF)
            {
                if (j == 0)
                {
                    status = 2;
                    break;
                }
                const int hl = Hcell(c0 + i, j - 1);
                em.emit(n, PG_OPC_I, 1);
                --j;
                if (sc == hl - PG_GAP_OPEN)
                {
                    sc += PG_GAP_OPEN;
                    inF = false;
                }
                else
                    sc += PG_GAP_EXT;
                continue;
            }
            const uint32_t rch = (uint8_t)refc[c0 + i];
            const uint32_t qch = qchar(j);
            const int sub = sub_score(nt_code(rch), nt_code(qch));
            if (i > 0 && j > 0)
            {
                if (sc == Hcell(c0 + i - 1, j - 1) + sub)
                {
                    const uint32_t op = (rch == 'N' || qch == 'N') ? PG_OPC_N : (rch == qch ? PG_OPC_M : PG_OPC_X);
                    em.emit(n, op, 1);
                        sc -= sub;
                    --i;
                    --j;
                    continue;
                }
            }
            else if (sc == sub)
            {
                // alignment start on the first row / column: never an 'X' (gssw.c:1655-1690)
                if (rch == 'N' || qch == 'N')
                {
                    em.emit(n, PG_OPC_N, 1);
                    }
                else if (rch == qch)
                {
                    em.emit(n, PG_OPC_M, 1);
                    }
                sc -= sub;
                --i;
                --j;
                continue;
            }
            if (j > 0)
            {
                // score == F(i,j) ?
                bool isF = false;
                const int kmax = (j + 1 - sc - PG_GAP_OPEN + PG_GAP_EXT) / (1 + PG_GAP_EXT);
                for (int kk = 1; kk <= kmax && kk <= j; ++kk)
                {
                    if (Hcell(c0 + i, j - kk) - PG_GAP_OPEN - (kk - 1) * PG_GAP_EXT == sc)
                    {
                        isF = true;
                        break;
                    }
                }
                if (isF)
                {
                    inF = true;
                    continue;
                }
            }
            if (i > 0)
            {
                inE = true;  // the only remaining explanation of H(i,j)
                continue;
            }
            // first column: E(0,j) is the seed, max over predecessors of their next-column E
            {
                int se = 0;
                for (uint32_t p = 0; p < nd.n_pred; ++p)
                {
                    const int en = seedE(a.preds[nd.pred_off + p], j);
                    se = en > se ? en : se;
                }
                if (sc == se)
                {
                    inE = true;
                    continue;
                }
            }
            break;  // try a diagonal into a predecessor
        }
        if (status != 0)
            break;
        if (sc != 0 && i > 0)
        {
            status = 2;
            break;
        }
        if (sc == 0)
        {
            if (j > -1)
            {
                em.emit(n, PG_OPC_S, (uint32_t)(j + 1));
                clipped += (uint32_t)(j + 1);
            }
            if (!em.node_has_ops)
                em.emit(n, PG_OPC_EMPTY, 0);  // gssw would print "n[]"
            break;
        }
        // ---- cross into a predecessor: first one (ascending id) consistent with diagonal / gap open
        //      / gap extend (gssw.c:2966-3161) --------------------------------------------------------
        int best_prev = -1;
        for (uint32_t p = 0; p < nd.n_pred && best_prev < 0; ++p)
        {
            const uint32_t pid = a.preds[nd.pred_off + p];
            if (!inE)
            {
                if (j < 1)
                    continue;
                const uint32_t rch = (uint8_t)refc[c0 + i];
                const uint32_t qch = qchar(j);
                const int sub = sub_score(nt_code(rch), nt_code(qch));
                const int hp = seedH(pid, j - 1);
                if (sc == hp + sub)
                {
                    const uint32_t op = (rch == 'N' || qch == 'N') ? PG_OPC_N : (rch == qch ? PG_OPC_M : PG_OPC_X);
                    em.emit(n, op, 1);
                        sc -= sub;
                    --j;
                    best_prev = (int)pid;
                }
            }
            else
            {
                const int hp = seedH(pid, j);
                const int en = seedE(pid, j);
                if (sc == hp - PG_GAP_OPEN)
                {
                    em.emit(n, PG_OPC_D, 1);
                        sc += PG_GAP_OPEN;
                    inE = false;
                    best_prev = (int)pid;
                }
                else if (sc == en)
                {
                    // en = max(E_p - ge, H_p - go) and the open test failed  =>  score == E_p - ge
                    em.emit(n, PG_OPC_D, 1);
                        sc += PG_GAP_EXT;
                    best_prev = (int)pid;
                }
            }
        }
        if (best_prev < 0)
        {
            status = 2;  // "Could not find a valid previous node": the reference asserts
            break;
        }
        if (!em.node_has_ops)
            em.emit(n, PG_OPC_EMPTY, 0);
        em.node_has_ops = false;
        n = (uint32_t)best_prev;
        i = (int)nodes[n].len - 1;
        if (sc <= 0)
            break;  // gssw.c:2778: the loop ends before the predecessor is visited
    }
    em.flush();
    if (em.overflow)
        status = 2;

    res.graph_pos = i + 1 < 0 ? 0 : i + 1;
    res.clipped = (uint16_t)clipped;
    res.status = (uint16_t)status;
    res.n_ops = (uint16_t)em.n;
    // compact: bump-allocate and copy the tail-aligned scratch slot into forward order
    const unsigned long long base = atomicAdd(a.ops_counter, (unsigned long long)em.n);
    for (uint32_t e = 0; e < em.n; ++e)
        a.ops[base + e] = em.slot[em.cap - em.n + e];
    res.ops_off = (uint32_t)base;
    a.results[ridx] = res;
}

hipError_t pg_launch_trace(const PgTraceArgs& args, hipStream_t stream)
{
    const uint32_t threads = args.n_pairs * PG_GROUPS;
    if (threads == 0)
        return hipSuccess;
    const dim3 grid((threads + 63) / 64), block(64);
    switch (args.C)
    {
    case 2: hipLaunchKernelGGL((pg_trace_kernel<2, false>), grid, block, 0, stream, args); break;
    case 4: hipLaunchKernelGGL((pg_trace_kernel<4, false>), grid, block, 0, stream, args); break;
    case 6: hipLaunchKernelGGL((pg_trace_kernel<6, false>), grid, block, 0, stream, args); break;
    case 8: hipLaunchKernelGGL((pg_trace_kernel<8, false>), grid, block, 0, stream, args); break;
    case 10: hipLaunchKernelGGL((pg_trace_kernel<10, false>), grid, block, 0, stream, args); break;
    case 12: hipLaunchKernelGGL((pg_trace_kernel<12, false>), grid, block, 0, stream, args); break;
    case 14: hipLaunchKernelGGL((pg_trace_kernel<14, false>), grid, block, 0, stream, args); break;
    case 16: hipLaunchKernelGGL((pg_trace_kernel<16, false>), grid, block, 0, stream, args); break;
    case PG_VAR_WIDE + 16: hipLaunchKernelGGL((pg_trace_kernel<16, true>), grid, block, 0, stream, args); break;
    case PG_VAR_WIDE + 20: hipLaunchKernelGGL((pg_trace_kernel<20, true>), grid, block, 0, stream, args); break;
    case PG_VAR_WIDE + 24: hipLaunchKernelGGL((pg_trace_kernel<24, true>), grid, block, 0, stream, args); break;
    case PG_VAR_WIDE + 28: hipLaunchKernelGGL((pg_trace_kernel<28, true>), grid, block, 0, stream, args); break;
    case PG_VAR_WIDE + 32: hipLaunchKernelGGL((pg_trace_kernel<32, true>), grid, block, 0, stream, args); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
