// pg_trace.hip -- strand pick + traceback kernel: one 16-lane DPP row per read (4 reads per wavefront, the fill kernel's
// own grouping).  The walk itself is sequential, but its cost is the latency of dependent single-byte loads from the H
// trace; the 16 lanes of a read take that latency 16 cells at a time:
//   * diagonal runs: lane d loads H(i-1-d, j-1-d) and the two characters of depth d, every lane checks its own step
//     (H(d-1) == H(d) + s(d)), one ballot gives the length of the run, and the run's M / X / N segments are emitted
//     run-length encoded -- one round of loads per 16 matched bases instead of 16, the loads of four rounds in flight together;
//   * the "score == F(i,j)" scan after a failed diagonal: 16 candidate gap lengths per round;
//   * everything else (gap steps, node boundaries) is the scalar logic, executed redundantly by the row's lanes (the same
//     address in every lane: one transaction).
// CIGAR elements are collected in a scratch slot of the workspace (written by the row's first lane) and copied out by all 16.
// The fill hands over the end cell as (graph column, lane); its row is found here (one round of loads).
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
#include <stdlib.h>

#include <algorithm>

#include "pg_device.h"
#include "pg_kernels.h"

// tuning constants: rounds of diagonal loads in flight together, and the register budget as wavefronts per SIMD
#ifndef PG_TRACE_BURST
#define PG_TRACE_BURST 4
#endif
#ifndef PG_TRACE_WPE
#define PG_TRACE_WPE 6
#endif

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
    uint32_t* slot;  // in the work item's scratch (workspace), written from the tail backwards by the row's first lane
    bool writer;
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
                if (writer)
                    slot[cap - 1 - n] = PG_OP_MAKE(last_node, last_op, last_len);
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

// GL = lanes per read in the FILL that wrote the trace (16, or 32 for the wide variants: two fill wavefronts per work item,
// reads 0-1 / 2-3, each with its own half of the item's trace and seed regions); this kernel walks with 16 lanes per read either way
// LEAN (pg_launch_trace_lean; byte variants): the reversed-graph fills of the pair's reads are in its reversed work item as always,
// the forward-graph fills are INSTANCES (PgInstItem: eight (read, strand) fills per wavefront) -- every read's better strand X, and
// its other strand Y where the reversed-graph fills already say that X is not unique and Y may be.  GraphAligner.cpp:340-356 with
// X = the strand with the larger score (the forward strand on a tie; a fill's best score is the same on the graph and on the
// reversed graph -- an alignment read backwards is an alignment of the reversed read to the reversed graph -- and the record is
// marked inconsistent if the forward fill of X says otherwise):
//   X unique                          -> X is returned whatever Y is (both unique: the larger score; Y not unique: the unique one)
//   X not unique, Y multi on the reversed graph -> neither is unique: the larger score, X
//   X not unique, Y not multi there   -> Y's forward fill decides (Y unique: Y, else X); not run yet (X's own forward fill was what
//                                        made it non-unique): it is queued here and the read comes back in pg_trace_lean2_kernel
template <int C, bool WIDE, int GL, bool LEAN = false> __device__ __forceinline__ void pg_trace_pair(const PgTraceArgs& a, const uint32_t pair, const uint32_t slot_of_row = PG_NONE)
{
    static_assert(!LEAN || (!WIDE && GL == PG_GROUP_LANES), "lean pass: byte variants");
    // No LDS and 80 VGPRs: this kernel runs on the second stream UNDER the next chunk's fill, whose 16 wavefronts per CU take
    // all 160 KB of LDS and four times 120 of the 512 VGPRs of a SIMD lane -- a traceback wavefront gets a place when a fill
    // wavefront retires and holds it for its grid-stride loop (pg_trace_blocks: how many do).
    const uint32_t lane = threadIdx.x;
    const uint32_t row = lane >> 4;  // this lane's 16-lane row (the row-level collectives' unit)
    // the read of the row: read `grp` of the pair -- its own number, or (the lean stage's second look at the reads its first left
    // open: every row has a (pair, read) of its own) the one it is given
    const uint32_t grp = slot_of_row == PG_NONE ? row : slot_of_row;
    const uint32_t k = lane & 15u;
    // row-level collectives; the control flow below is uniform within a row, so a lane only ever talks to active lanes
    auto row_ballot = [&](bool p) -> uint32_t { return (uint32_t)(__ballot(p) >> (row * 16u)) & 0xFFFFu; };
    auto row_get = [&](uint32_t v, uint32_t src) -> uint32_t { return (uint32_t)__shfl((int)v, (int)(row * 16u + src)); };
    const PgWorkItem* fw = a.items + 2 * (size_t)pair;
    const uint32_t ridx = fw->read[grp];
    if (ridx == PG_NONE)
        return;
    const bool writer = k == 0u;
    const uint32_t off = a.base_off[ridx];
    const int L = (int)(a.base_off[ridx + 1] - off);
    const char* __restrict__ bases = a.bases + off;

    const PgFillSummary* fsF = a.fillsum + ((size_t)(2 * pair) * PG_GROUPS + grp) * 2;
    const PgFillSummary* fsR = a.fillsum + ((size_t)(2 * pair + 1) * PG_GROUPS + grp) * 2;
    const bool both = (a.flags & PG_AF_BOTH_STRANDS) != 0;
    const bool revg = (a.flags & PG_AF_REVERSE_GRAPH) != 0;

    // ---- GraphAligner.cpp:340-356 ------------------------------------------------------------------
    int m0, m1, m2, m3, s;
    bool unique;
    bool return_reverse = false;
    PgFillSummary fs;
    int score_fwd_strand, score_rc_strand;
    uint32_t hsel = 0;           // which 16-bit half of the fill's registers (= byte pair of its trace dwords) the walked fill is
    uint32_t inst_group = grp;   // ... and which 16-lane group of its wavefront
    const PgInstItem* inp = nullptr;
    bool other_skipped = false, inconsistent = false;
    if (LEAN)
    {
        m2 = fsR[0].multi;
        m3 = fsR[1].multi;
        const int SA = fsR[0].score, SB = fsR[1].score;
        const int X = SA >= SB ? 0 : 1, Y = 1 - X;
        // the run of the pair (last segment whose pair_begin <= pair): instance item of X = run start + (rank in the run) / 2, half = its parity
        uint32_t lo = 0, hi = a.n_segments;
        while (hi - lo > 1)
        {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.segments[mid].pair_begin <= pair)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t rs = a.segments[lo].pair_begin;
        // (fused kernel: the couple's X strands sit in the LEADER pair's slot; launches of their own: item rs + rank / 2)
        const uint32_t qX = a.fused ? rs + ((pair - rs) & ~1u) : rs + (pair - rs) / 2u, hX = (pair - rs) & 1u;
        const PgFillSummary fX = a.fillsum[((size_t)(2 * qX) * PG_GROUPS + grp) * 2 + hX];
        const int mXf = fX.multi, mXr = X ? m3 : m2, mYr = X ? m2 : m3;
        inconsistent = fX.score != (X ? SB : SA);
        uint32_t yl = a.yloc[ridx];
        if (yl != PG_NONE && (yl & PG_YLOC_PENDING))
        {
            // the fused kernel queued the other strand's forward fill for the chunk's second forward launch and listed the read: the
            // first look passes it by, the second (slot_of_row given) finds the fill made
            if (slot_of_row == PG_NONE)
                return;
            yl &= ~PG_YLOC_PENDING;
        }
        int mYf = 0;
        PgFillSummary fY{};
        uint32_t qY = 0, gY = 0, hY = 0;
        if (yl != PG_NONE)
        {
            qY = yl >> 3;
            gY = (yl >> 1) & 3u;
            hY = yl & 1u;
            fY = a.fillsum[((size_t)(2 * qY) * PG_GROUPS + gY) * 2 + hY];
            mYf = fY.multi;
            inconsistent = inconsistent || fY.score != (X ? SA : SB);
        }
        other_skipped = yl == PG_NONE;
        const bool X_u = !mXf && !mXr;
        int ret = X;
        bool undecided = false;
        if (!X_u && !mYr)
        {
            if (yl == PG_NONE)
                undecided = true;
            else if (!mYf)
                ret = Y;
        }
        if (undecided && a.fused)
        {
            // (the fused kernel runs every forward fill a record needs before it ends: this cannot be)
            undecided = false;
            inconsistent = true;
        }
        if (undecided)
        {
            // X is not unique through its own forward fill and Y still may be: Y's forward fill is queued -- an instance behind
            // the run's others (the pick kernel's slot space: X instances take 4 per pair, the others follow, 8 per pair slot in all) --
            // and the read is listed for the stage's second look (pg_trace_lean2_kernel), which finds yloc set
            if (writer)
            {
                const PgPlanSegment sg = a.segments[lo];
                const uint32_t count = a.group_count[sg.group];
                uint32_t ne = 0;
                if (count > 4u * sg.first_pair)
                {
                    ne = (count - 4u * sg.first_pair + 3u) / 4u;
                    ne = ne < sg.n_pairs ? ne : sg.n_pairs;
                }
                const uint32_t e = atomicAdd(&a.extra[rs], 1u);
                const uint32_t t = 4u * ne + e;
                const uint32_t qq = rs + t / 8u, within = t & 7u;
                PgInstItem* it = a.inst_rw + qq;
                it->inst[within >> 2][within & 3u] = ridx | (Y ? PG_INST_RC : 0u);
                it->pad = 1u;  // (an item of the second forward launch)
                a.yloc_rw[ridx] = (qq << 3) | ((within & 3u) << 1) | (within >> 2);
                const uint32_t li = atomicAdd(a.ucount, 1u);
                a.ulist[li] = (pair << 2) | grp;
            }
            return;
        }
        s = ret;
        return_reverse = ret == 1;
        unique = ret == X ? X_u : true;  // (Y is only ever returned as the unique one)
        m0 = X == 0 ? mXf : mYf;
        m1 = X == 1 ? mXf : mYf;
        fs = ret == X ? fX : fY;
        score_fwd_strand = X == 0 ? fX.score : (yl != PG_NONE ? fY.score : SA);
        score_rc_strand = X == 1 ? fX.score : (yl != PG_NONE ? fY.score : SB);
        inp = a.inst + (ret == X ? qX : qY);
        hsel = ret == X ? hX : hY;
        inst_group = ret == X ? grp : gY;
    }
    else
    {
        m0 = fsF[0].multi;
        m1 = both ? fsF[1].multi : 0;
        m2 = revg ? fsR[0].multi : 0;
        m3 = (revg && both) ? fsR[1].multi : 0;
        const bool fwd_unique = !m0 && !m2;
        const bool rev_unique = !m1 && !m3;
        if (!fwd_unique && rev_unique && both)
            return_reverse = true;
        else if (fwd_unique && !rev_unique)
            return_reverse = false;
        else if (both)
            return_reverse = fsF[0].score < fsF[1].score;
        s = return_reverse ? 1 : 0;
        unique = return_reverse ? rev_unique : fwd_unique;
        fs = fsF[s];
        score_fwd_strand = fsF[0].score;
        score_rc_strand = both ? fsF[1].score : -1;
        hsel = (uint32_t)s;
    }

    pg_result res;
    res.graph_pos = 0;
    res.score = (int16_t)fs.score;
    res.mapq = unique ? 60 : 0;
    res.is_unique = unique ? 1 : 0;
    res.returned_reverse = return_reverse ? 1 : 0;
    res.multi_mask = (uint8_t)(m0 | (m1 << 1) | (m2 << 2) | (m3 << 3) | (LEAN && other_skipped ? PG_MULTI_OTHER_FWD_SKIPPED : 0));
    res.n_ops = 0;
    res.ops_off = 0;
    res.strand_score[0] = (int16_t)score_fwd_strand;
    res.strand_score[1] = (int16_t)score_rc_strand;
    res.clipped = 0;
    res.status = 0;
    if (LEAN && inconsistent)
    {
        // (a forward fill whose best score is not the reversed-graph fill's: the lean pick does not hold for this read)
        res.status = 2;
        if (writer)
            a.results[ridx] = res;
        return;
    }

    if (fs.score <= 0)
    {
        // all-zero fill: gssw returns an empty CIGAR at position 0 (gssw.c:2728-2732, 2778)
        res.status = 1;
        if (writer)
            a.results[ridx] = res;
        return;
    }

    const PgGraphDev gdev = a.graphs[fw->graph];
    const PgNode* __restrict__ nodes = a.nodes + gdev.dir[0].node_off;
    const char* __restrict__ refc = a.seqchars + gdev.seq_off;
    // where this read's lanes sit in the fill wavefront's records, and which half of the item's regions that wavefront owns
    constexpr uint32_t FILL_GROUPS = 64 / GL;
    const uint32_t half = LEAN ? 0u : grp / FILL_GROUPS, lane0 = LEAN ? inst_group * GL : (grp % FILL_GROUPS) * GL;
    const uint8_t* __restrict__ trace = a.workspace + (LEAN ? inp->trace_off : fw->trace_off)
        + (size_t)half * pg_fill_steps_lanes(gdev.dir[0].ncols, GL) * 64 * (2 * C);
    const uint32_t* __restrict__ seed = (const uint32_t*)(a.workspace + (LEAN ? inp->seed_off : fw->seed_off)) + (size_t)half * gdev.dir[0].n_nodes * 64 * (WIDE ? 2 * C : C);

    auto qchar = [&](int j) -> uint32_t {
        return s == 0 ? upper_c((uint8_t)bases[j]) : comp_c((uint8_t)bases[L - 1 - j]);
    };
    // byte variants: trace [step / 2][C/2][lane][step & 1] dwords of bytes (A_r, A_r+1, B_r, B_r+1), seed [node][lane][C] dwords of
    // bytes (H_A, H_B, Enext_A, Enext_B); wide variants: the same trace, seed [node][lane][2C] dwords (H_A | H_B << 16),
    // (Enext_A | Enext_B << 16), each 16-bit field = 0x6400 | score
    uint32_t n = (uint32_t)fs.max_node;  // the node being walked
    auto seedH = [&](uint32_t node, int j) -> int {
        const uint32_t kq = (uint32_t)j / C, r = (uint32_t)j % C;
        if (WIDE)
            return (int)((seed[((size_t)node * 64 + (lane0 + kq)) * (2 * C) + 2 * r] >> (16 * s)) & 0x3FFu);
        return (int)((seed[((size_t)node * 64 + (lane0 + kq)) * C + r] >> (8 * hsel)) & 0xFFu);
    };
    auto seedE = [&](uint32_t node, int j) -> int {
        const uint32_t kq = (uint32_t)j / C, r = (uint32_t)j % C;
        if (WIDE)
            return (int)((seed[((size_t)node * 64 + (lane0 + kq)) * (2 * C) + 2 * r + 1] >> (16 * s)) & 0x3FFu);
        return (int)((seed[((size_t)node * 64 + (lane0 + kq)) * C + r] >> (16 + 8 * hsel)) & 0xFFu);
    };

    // A trace dword holds diagonal neighbours (pg_fill.hip, column()): the odd row r of the step it was written in and the even
    // row r - 1 of the step before -- so an even row is found one step on.  The even rows of a node's LAST column are not in
    // the trace (the step after it starts another node); they are in that node's seed.  `at_end`: col is the last column of the
    // node being walked (only the F tests look at the walk's own column; diagonals and E look at the one before it).
    auto Hcell = [&](uint32_t col, int j, bool at_end) -> int {
        const uint32_t kq = (uint32_t)j / C, r = (uint32_t)j % C;
        // the fill kernel stores (score + tau) of the step the cell was computed in (PG_TAU0, pg_device.h)
        const uint32_t tau = PG_TAU0 + ((col + kq) & 255u);
        if (!(r & 1u) && at_end)
            return seedH(n, j) & (WIDE ? 0xFF : 0x3FF);
        // ([step / 2][C / 2][lane][step & 1] dwords: the fill stores an even step and the odd one behind it together)
        const uint32_t step = col + kq + ((r & 1u) ^ 1u);
        const size_t dw = (((size_t)(step >> 1) * (C / 2) + r / 2) * 64 + (lane0 + kq)) * 2 + (step & 1u);
        // (non-temporal: a trace byte is read once, and what the walk pulls through the L2 competes with the next chunk's fill)
        return (int)(((uint32_t)__builtin_nontemporal_load(&trace[dw * 4 + (r & 1u) + 2u * hsel]) - tau) & 0xFFu);
    };
    // The byte is the whole score in the byte variants (reads <= 250 bases).  In the wide ones it is the score modulo 256: the
    // walk itself carries the exact score (`sc`), and every test below relates the scores of cells that are neighbours or on
    // one short diagonal / vertical run, whose true values differ by far less than 128 -- so equality modulo 256 is equality
    // (same(): x == y for the byte variants, x == y mod 256 for the wide ones); where a run is long (the F scan) the exact
    // values are carried from cell to cell by the signed 8-bit differences of the bytes.
    auto same = [&](int x, int y) -> bool { return WIDE ? (((uint32_t)(x - y)) & 0xFFu) == 0u : x == y; };
    // inclusive prefix sum over the 16 lanes of the row
    auto row_scan = [&](int v) -> int {
#pragma unroll
        for (uint32_t off = 1; off < 16u; off *= 2u)
        {
            const int up = (int)row_get((uint32_t)v, k >= off ? k - off : 0u);
            if (k >= off)
                v += up;
        }
        return v;
    };
    Emitter em;
    em.cap = pg_ops_cap(WIDE ? PG_VAR_WIDE + C * (GL / 16) : C);
    em.slot = (uint32_t*)(a.workspace_rw + a.items[2 * pair + 1].trace_off) + grp * em.cap;  // [4 reads][pg_ops_cap]
    em.writer = writer;
    em.n = 0;
    em.last_op = 0xFFu;
    em.last_node = 0;
    em.last_len = 0;
    em.overflow = false;
    em.node_has_ops = false;

    // The fill leaves the end cell as (graph column, first row of the fill lane that holds it): the node-local column follows
    // from the node, the row = the first of that lane's C rows holding the score in that column (gssw's read_end1, the
    // smallest read index in the column: lanes are ordered by the key, rows here) -- one round of loads for the row's lanes.
    PgNode nd = nodes[n];
    int i = fs.end_col - (int)nd.col_start;
    int j = fs.read_end;
    {
        const bool at_end = i == (int)nd.len - 1;
        uint32_t hit = 0u;  // bit r: row j + r holds the score
#pragma unroll
        for (int base = 0; base < C; base += 16)
        {
            const int r = base + (int)k;
            bool eq = false;
            if (r < C)
            {
                const int h = Hcell((uint32_t)fs.end_col, j + r, at_end);
                eq = WIDE ? ((uint32_t)(h ^ fs.score) & 0xFFu) == 0u : h == fs.score;
            }
            hit |= row_ballot(eq) << base;
        }
        if (hit)
            j += (int)__builtin_ctz(hit);
    }
    int sc = fs.score;
    bool inE = false, inF = false;
    bool skip_probe = false;
    int status = 0;
    uint32_t clipped = 0;
    if (L - 1 - j > 0)
    {
        em.emit(n, PG_OPC_S, (uint32_t)(L - 1 - j));
        clipped += (uint32_t)(L - 1 - j);
    }

    for (int guard = 0; guard < 0x7fffffff; ++guard)
    {
        const uint32_t c0 = nd.col_start;
        // ---- within-node traceback (gssw.c:1214-1808) ---------------------------------------------
        while (sc > 0 && i >= 0 && j >= 0)
        {
            if (inE)
            {
                if (i == 0)
                    break;
                const int hup = Hcell(c0 + i - 1, j, false);
                em.emit(n, PG_OPC_D, 1);
                --i;
                if (same(sc, hup - PG_GAP_OPEN))
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
                const int hl = Hcell(c0 + i, j - 1, i == (int)nd.len - 1);
                em.emit(n, PG_OPC_I, 1);
                --j;
                if (same(sc, hl - PG_GAP_OPEN))
                {
                    sc += PG_GAP_OPEN;
                    inF = false;
                }
                else
                    sc += PG_GAP_EXT;
                continue;
            }
            if (i > 0 && j > 0 && !skip_probe)
            {
                // ---- diagonal run, 16 cells per round: lane d looks at cell (i - d, j - d) -----------------------------
                // The loads of PG_TRACE_BURST rounds go out together (a round's addresses do not depend on the round before it,
                // only on its being a full run): under the next chunk's fill a dependent load takes microseconds, and the walk is
                // a chain of them.  A round after a broken run is never looked at; its loads are the price.
                int hdv[PG_TRACE_BURST];
                uint32_t rcv[PG_TRACE_BURST], qcv[PG_TRACE_BURST];
#pragma unroll
                for (int rd = 0; rd < PG_TRACE_BURST; ++rd)
                {
                    const int ii = i - (int)k - 16 * rd, jj = j - (int)k - 16 * rd;
                    hdv[rd] = 0;
                    rcv[rd] = 0;
                    qcv[rd] = 0;
                    if (ii > 0 && jj > 0)
                    {
                        hdv[rd] = Hcell(c0 + ii - 1, jj - 1, false);
                        rcv[rd] = (uint8_t)refc[c0 + ii];
                        qcv[rd] = qchar(jj);
                    }
                }
                bool progressed = false;
#pragma unroll
                for (int rd = 0; rd < PG_TRACE_BURST; ++rd)
                {
                    if (rd > 0 && !(sc > 0 && i > 0 && j > 0))
                        break;  // what the loop head would decide after a full run
                    const int ii = i - (int)k, jj = j - (int)k;
                    const bool valid = ii > 0 && jj > 0;
                    const int hd = hdv[rd];
                    int subd = 0;
                    uint32_t opd = PG_OPC_M;
                    if (valid)
                    {
                        const uint32_t rc = rcv[rd], qc = qcv[rd];
                        subd = sub_score(nt_code(rc), nt_code(qc));
                        opd = (rc == 'N' || qc == 'N') ? PG_OPC_N : (rc == qc ? PG_OPC_M : PG_OPC_X);
                    }
                    // the score the walk holds when it arrives at depth d: H of depth d - 1 (byte variants), or the exact score
                    // minus the substitution scores of the depths before (wide variants, where H is known modulo 256 only)
                    int cur;
                    int sub_before = 0;
                    if (WIDE)
                    {
                        sub_before = row_scan(subd) - subd;
                        cur = sc - sub_before;
                    }
                    else
                    {
                        const int above = (int)row_get((uint32_t)hd, k == 0u ? 0u : k - 1u);
                        cur = k == 0u ? sc : above;
                    }
                    const bool ok = valid && cur > 0 && same(cur, hd + subd);
                    const uint32_t fail = ~row_ballot(ok) & 0xFFFFu;
                    const uint32_t run = fail ? (uint32_t)__builtin_ctz(fail) : 16u;
                    if (run > 0u)
                    {
                        const uint32_t prev_op = row_get(opd, k == 0u ? 0u : k - 1u);
                        uint32_t starts = row_ballot(k < run && (k == 0u || opd != prev_op));
                        while (starts)
                        {
                            const uint32_t s0 = (uint32_t)__builtin_ctz(starts);
                            starts &= starts - 1u;
                            const uint32_t s1 = starts ? (uint32_t)__builtin_ctz(starts) : run;
                            em.emit(n, row_get(opd, s0), s1 - s0);
                        }
                        if (WIDE)
                            sc -= (int)row_get((uint32_t)(sub_before + subd), run - 1u);
                        else
                            sc = (int)row_get((uint32_t)hd, run - 1u);
                        i -= (int)run;
                        j -= (int)run;
                        progressed = true;
                    }
                    skip_probe = run < 16u;  // the cell the run stopped at is handled by the scalar logic below
                    if (run < 16u)
                        break;
                }
                if (progressed)
                    continue;
            }
            skip_probe = false;
            const uint32_t rch = (uint8_t)refc[c0 + i];
            const uint32_t qch = qchar(j);
            const int sub = sub_score(nt_code(rch), nt_code(qch));
            if (i > 0 && j > 0)
            {
                if (same(sc, Hcell(c0 + i - 1, j - 1, false) + sub))
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
                const int lim = kmax < j ? kmax : j;
                int run_exact = sc;                        // exact H(i, j - (base - 1)); H(i, j) = sc
                uint32_t run_byte = (uint32_t)sc & 0xFFu;  // ... and its byte
                for (int base = 1; base <= lim && !isF; base += 16)
                {  // 16 gap lengths per round
                    const int kk = base + (int)k;
                    const bool in = kk <= lim;
                    int h = 0;
                    if (in)
                        h = Hcell(c0 + i, j - kk, i == (int)nd.len - 1);
                    if (WIDE)
                    {
                        // exact values along the column: add up the signed differences of neighbouring bytes
                        const uint32_t above = row_get((uint32_t)h, k == 0u ? 0u : k - 1u);
                        const int diff = (int)(int8_t)(uint8_t)((uint32_t)h - (k == 0u ? run_byte : above));
                        const int exact = run_exact + row_scan(in ? diff : 0);
                        run_exact = (int)row_get((uint32_t)exact, 15u);
                        run_byte = row_get((uint32_t)h, 15u);
                        h = exact;
                    }
                    const bool hit = in && h - PG_GAP_OPEN - (kk - 1) * PG_GAP_EXT == sc;
                    isF = row_ballot(hit) != 0u;
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
        nd = nodes[n];
        i = (int)nd.len - 1;
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
    // compact: bump-allocate (first lane) and copy the tail-aligned slot into forward order (all 16 lanes)
    __threadfence_block();
    unsigned long long base = 0;
    if (writer)
        base = atomicAdd(a.ops_counter, (unsigned long long)em.n);
    base = ((unsigned long long)row_get((uint32_t)(base >> 32), 0u) << 32) | row_get((uint32_t)base, 0u);
    for (uint32_t e = k; e < em.n; e += 16u)
        a.ops[base + e] = em.slot[em.cap - em.n + e];
    res.ops_off = (uint32_t)base;
    if (writer)
        a.results[ridx] = res;
}

// One wavefront per workgroup, walking work-item pairs blockIdx.x, blockIdx.x + gridDim.x, ...: the launcher bounds the number
// of wavefronts in flight (pg_trace_blocks below).
template <int C, bool WIDE, int GL = PG_GROUP_LANES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PG_TRACE_WPE, PG_TRACE_WPE))) void pg_trace_kernel(PgTraceArgs a)
{
    for (uint32_t p = blockIdx.x; p < a.n_pairs; p += gridDim.x)
        pg_trace_pair<C, WIDE, GL>(a, a.pair_begin + p);
}

template <int C> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PG_TRACE_WPE, PG_TRACE_WPE))) void pg_trace_lean_kernel(PgTraceArgs a)
{
    for (uint32_t p = blockIdx.x; p < a.n_pairs; p += gridDim.x)
        pg_trace_pair<C, false, PG_GROUP_LANES, true>(a, a.pair_begin + p);
}

// the lean stage's second look: the reads its first one listed (their other strand's forward fill has run since), one per 16-lane row
template <int C> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PG_TRACE_WPE, PG_TRACE_WPE))) void pg_trace_lean2_kernel(PgTraceArgs a)
{
    const uint32_t count = *a.ucount;
    for (uint32_t w = blockIdx.x; w * 4u < count; w += gridDim.x)
    {
        const uint32_t idx = w * 4u + (threadIdx.x >> 4);
        if (idx < count)
        {
            const uint32_t ent = a.ulist[idx];
            pg_trace_pair<C, false, PG_GROUP_LANES, true>(a, ent >> 2, ent & 3u);
        }
    }
}

// Wavefronts of one traceback launch.  The walk is a chain of dependent loads that each pull a whole 128-byte line for a few
// bytes (profiles/r03_sector_probe.json), and it runs under the next chunk's fill, which writes 3.5 TB/s the whole time.
// Measured with stand-in kernels in the walk's place (profiles/r03_trace_tax.md): wavefronts that only sit beside the fill
// cost it little; dependent random-line reads cost it in proportion to their bytes.  A traceback wavefront (80 registers)
// does take the place of a fill wavefront on its SIMD (4 x 120 of the 512 registers are the fill's), so the number in flight
// trades the time the traceback holds those places against how many it holds: measured on the round-4 kernels
// (profiles/r04_trace_blocks_ab.jsonl, three boxes) 4 per CU -6 %, 8 per CU the old default, 11-12 per CU +1.2 ... 1.4 %, 16
// and more +0.2 ... 0.4 % (PG_TRACE_BLOCKS overrides; 0 = one per pair).
static uint32_t pg_trace_blocks(uint32_t n_pairs)
{
    static const long cap = [] {
        const char* e = getenv("PG_TRACE_BLOCKS");
        if (e)
            return atol(e);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 12L * cus;
    }();
    return cap > 0 && (uint32_t)cap < n_pairs ? (uint32_t)cap : n_pairs;
}

template <int C, bool WIDE, int GL = PG_GROUP_LANES> static hipError_t launch_trace_c(const PgTraceArgs& args, hipStream_t stream)
{
    hipLaunchKernelGGL((pg_trace_kernel<C, WIDE, GL>), dim3(pg_trace_blocks(args.n_pairs)), dim3(64), 0, stream, args);
    return hipGetLastError();
}

template <int C> static hipError_t launch_trace_lean_c(const PgTraceArgs& args, hipStream_t stream)
{
    hipLaunchKernelGGL((pg_trace_lean_kernel<C>), dim3(pg_trace_blocks(args.n_pairs)), dim3(64), 0, stream, args);
    return hipGetLastError();
}

template <int C> static hipError_t launch_trace_lean2_c(const PgTraceArgs& args, hipStream_t stream)
{
    // (the list is a few per cent of the chunk's reads at most: a bounded grid, the kernel strides over what the counter says)
    const uint32_t blocks = std::min<uint32_t>(pg_trace_blocks(args.n_pairs), 4096u);
    hipLaunchKernelGGL((pg_trace_lean2_kernel<C>), dim3(blocks ? blocks : 1u), dim3(64), 0, stream, args);
    return hipGetLastError();
}

hipError_t pg_launch_trace_lean2(const PgTraceArgs& args, hipStream_t stream)
{
    if (args.n_pairs == 0)
        return hipSuccess;
    switch (args.C)
    {
    case 2: return launch_trace_lean2_c<2>(args, stream);
    case 4: return launch_trace_lean2_c<4>(args, stream);
    case 6: return launch_trace_lean2_c<6>(args, stream);
    case 8: return launch_trace_lean2_c<8>(args, stream);
    case 10: return launch_trace_lean2_c<10>(args, stream);
    case 12: return launch_trace_lean2_c<12>(args, stream);
    case 14: return launch_trace_lean2_c<14>(args, stream);
    case 16: return launch_trace_lean2_c<16>(args, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t pg_launch_trace_lean(const PgTraceArgs& args, hipStream_t stream)
{
    if (args.n_pairs == 0)
        return hipSuccess;
    switch (args.C)
    {
    case 2: return launch_trace_lean_c<2>(args, stream);
    case 4: return launch_trace_lean_c<4>(args, stream);
    case 6: return launch_trace_lean_c<6>(args, stream);
    case 8: return launch_trace_lean_c<8>(args, stream);
    case 10: return launch_trace_lean_c<10>(args, stream);
    case 12: return launch_trace_lean_c<12>(args, stream);
    case 14: return launch_trace_lean_c<14>(args, stream);
    case 16: return launch_trace_lean_c<16>(args, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t pg_launch_trace(const PgTraceArgs& args, hipStream_t stream)
{
    if (args.n_pairs == 0)
        return hipSuccess;
    if (pg_var_wide(args.C) && args.wide32)
    {
        switch (pg_var_c(args.C))
        {
        case 16: return launch_trace_c<8, true, 32>(args, stream);
        case 20: return launch_trace_c<10, true, 32>(args, stream);
        case 24: return launch_trace_c<12, true, 32>(args, stream);
        case 28: return launch_trace_c<14, true, 32>(args, stream);
        case 32: return launch_trace_c<16, true, 32>(args, stream);
        default: return hipErrorInvalidValue;
        }
    }
    switch (args.C)
    {
    case 2: return launch_trace_c<2, false>(args, stream);
    case 4: return launch_trace_c<4, false>(args, stream);
    case 6: return launch_trace_c<6, false>(args, stream);
    case 8: return launch_trace_c<8, false>(args, stream);
    case 10: return launch_trace_c<10, false>(args, stream);
    case 12: return launch_trace_c<12, false>(args, stream);
    case 14: return launch_trace_c<14, false>(args, stream);
    case 16: return launch_trace_c<16, false>(args, stream);
    case PG_VAR_WIDE + 16: return launch_trace_c<16, true>(args, stream);
    case PG_VAR_WIDE + 20: return launch_trace_c<20, true>(args, stream);
    case PG_VAR_WIDE + 24: return launch_trace_c<24, true>(args, stream);
    case PG_VAR_WIDE + 28: return launch_trace_c<28, true>(args, stream);
    case PG_VAR_WIDE + 32: return launch_trace_c<32, true>(args, stream);
    default: return hipErrorInvalidValue;
    }
}
