// pg_fill.hip -- the DP fill kernel (gfx950 / CDNA4, hand-written HIP).
//
// Replaces, for a whole batch at once, what the reference does per read in
//   gssw_graph_fill           external/gssw/gssw.c:3963-4044   (topological node loop, max_node rule)
//   gssw_create_seed_byte     gssw.c:3897-3931                 (lane-wise max over predecessor seeds)
//   gssw_sw_sse2_byte         gssw.c:153-473                   (affine-gap fill of one node)
//   gssw_qP_byte              gssw.c:72-98                     (query profile)
//   alignsEndAtMultNodes      src/c++/lib/grm/GraphAligner.cpp:170-212 (fused: per-node maxima instead of a rescan)
//
// Mapping onto CDNA4
//   * one 64-lane wavefront = 4 reads x 16 lanes; every 16-lane DPP row owns one read.  Lane k of a row
//     owns read rows [k*C, (k+1)*C) (C = ceil(L/32)*2, template parameter) and sweeps the graph's
//     columns node by node in topological order, skewed by one column per lane (anti-diagonal
//     wavefront): at step t lane k works on column t-k.
//   * both strands of a read (read and its reverse complement) run in the two 16-bit halves of every
//     VGPR, so one wavefront performs 8 fills at once.  A score n is held as the f16 number 1024 + n in a
//     frame that moves by one per step (bit pattern 0x6400 + n + tau: every integer below 2048 is exact
//     in f16, so this is integer arithmetic in disguise): the three maxima of a cell are
//     v_pk_maximum3_f16 / v_pk_max_u16 on those patterns (gfx950 has a three-input packed maximum for
//     f16 and none for integers), the three additions are plain 32-bit v_add_u32 on the patterns of both
//     halves at once (two issue cycles where a packed add takes four; pg_pk16.h).  Results equal gssw's
//     saturating u8 / i16 arithmetic bit for bit.  No MFMA: this is integer DP, not a contraction.
//   * the only cross-lane traffic is (H of the row above on the previous column, running F) handed to
//     the next lane with two row_shr:1 DPP moves per step; the column's meta word (base code, node
//     boundary flags) rides the same shift, lane 0 reads it with a scalar load.
//   * the query profile of the 8 fills lives in LDS ([read][ref code][row] packed dwords); each step
//     costs C/2 ds_read_b64 per lane.
//   * H of every cell of the forward-graph fills is written to HBM as bytes, laid out by pipeline
//     step so that one step of one wavefront is a single contiguous 64*2C-byte store (a dword holds
//     diagonal neighbours: the odd row of this step, the even row of the step before); the traceback
//     kernel re-derives E/F decisions from H (see pg_trace.hip).
//   * node boundaries: the last column (H, next-column E) of a node is stored once per lane in a
//     per-wavefront seed region; a node's first column takes the lane-wise max over its predecessors'
//     seeds (the predecessor that directly precedes it in the layout stays in registers, the seeds a far
//     successor wants in a two-entry register cache).
//   * wherever a wavefront waits for memory it pays the latency of a chip that moves 3.5 TB/s of its
//     own stores (one in-order counter): no load in the step loop on the usual graphs, one round trip
//     in the tail (profiles/r03_trace_tax.md).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <stdlib.h>

#include "pg_device.h"
#include "pg_kernels.h"

#include "pg_pk16.h"

__device__ __forceinline__ uint32_t nt_code(uint32_t c)
{  // gssw.c:4206-4220
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
{  // graph-tools SequenceOperations.cpp:66-81: anything but upper-case ACGT becomes 'N'
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
{  // gssw.c:4188-4204 (match 1, mismatch 4)
    return (a == 4u || b == 4u) ? 0 : (a == b ? 1 : -4);
}

// column meta words are fetched this many steps ahead of their use (scalar loads; a step is ~0.6 us of wall clock with four
// wavefronts per SIMD, a scalar-cache miss an L2 round trip under the chip's full traffic)
#ifndef PG_META_AHEAD
#define PG_META_AHEAD 2
#endif

// INST (lean forward pass, byte variants, DIR 0): the wavefront's eight fills are eight (read, strand) instances of a PgInstItem --
// the two halves of a register belong to two different reads -- instead of the two strands of four reads
template <int C, int DIR, bool WIDE, int GL = PG_GROUP_LANES, bool INST = false>
__device__ __forceinline__ void pg_fill_body(const PgFillArgs& a, uint32_t pair, uint32_t* lds, uint32_t half = 0)
{
    static_assert(!INST || (DIR == 0 && !WIDE && GL == PG_GROUP_LANES), "instance items: forward graph, byte variants");
    // GL lanes per read: 16 = the four reads of a work item in one wavefront; 32 (wide variants) = two reads per wavefront,
    // wavefront `half` of the item takes reads 2 * half, 2 * half + 1 and its own half of the item's trace / seed regions
    constexpr int GROUPS = 64 / GL;
    constexpr int PAD = WIDE ? PG_PAD_SCORE_WIDE : PG_PAD_SCORE;
    // Register / LDS budget.  The byte variants keep the next column's profile rows (fetched one step ahead) and the lane's
    // last seed in registers: 128 VGPRs or fewer up to C = 12 (4 wavefronts per SIMD; the 4 KB x C / 4 of LDS profile allow
    // 16 / 13 wavefronts per CU at C = 10 / 12) and at most 170 up to C = 16 (3 per SIMD; LDS allows 10 per CU there).  The
    // wide variants from C = 20 on would fall to one wavefront per SIMD with the prefetch registers; they fetch the rows at
    // the start of the step instead.
    constexpr bool PREFETCH = WIDE ? C <= 16 : true;
    constexpr bool SEEDCACHE = !WIDE || (GL == 32 && C <= 14);   // (16 rows x 32 lanes: 174 registers, two wavefronts per SIMD)
    constexpr bool SEEDCACHE2 = SEEDCACHE && (WIDE ? C <= 8 : C <= 12);
    constexpr int TRACE_DW = C / 2;             // dwords of H trace per lane per step: one byte per cell, two strands
    constexpr int SEED_DW = WIDE ? 2 * C : C;   // dwords of seed per lane per node
    constexpr int ROWS = GL * C;
    // LDS holds the profiles of the four real reference codes only: [GROUPS reads][4 codes][ROWS] packed (strand A | strand B << 16)
    // = 10 240 B at C = 10, i.e. 16 wavefronts per CU (the kernel is latency-sensitive: 12 -> 16 waves is worth ~8 %).  Code 4
    // (N / idle column) scores 0 on real rows and PAD on padding rows: synthesised in registers on the rare columns that
    // carry it.  The per-node maxima keys live in the workspace behind this item's seed region (global atomics, 3 per lane
    // per sweep) for the same reason.
    uint32_t* prof = lds;

    const int lane = threadIdx.x;
    const int lgrp = lane / GL;                    // read of this wavefront the lane works for
    const int grp = (int)half * GROUPS + lgrp;     // ... = read slot of the work item
    const int k = lane % GL;

    // items come in (forward graph, reversed graph) pairs: this instantiation takes the DIR member
    const uint32_t item_idx = a.item_begin + 2 * pair + DIR;
    const PgWorkItem* itp = a.items + item_idx;
    const PgInstItem* inp = INST ? a.inst + (a.item_begin / 2 + pair) : nullptr;
    // (the fused lean kernel's wavefront wrote the item itself a moment ago: device-scope loads, not the CU's vector cache)
    auto inst_at = [&](int h, int g) -> uint32_t { return __hip_atomic_load(&inp->inst[h][g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // an EMPTY slot of a plan re-written by the cascade's hand-over (pg_batch_retire_mapped: a group's active reads come first,
    // so the wavefront's first read says it all): nothing to fill, nothing the traceback will look at
    // (instance items are filled from (half 0, group 0) on)
    if ((INST ? inst_at(0, 0) : itp->read[(int)half * GROUPS]) == PG_NONE)
        return;
    if (INST && a.both_dirs == 4u && __hip_atomic_load(&inp->pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        return;  // (the lean stage's second forward launch: only the items the traceback's first look added instances to)
    const uint32_t graph = INST ? __hip_atomic_load(&inp->graph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : itp->graph;
    const uint64_t item_seed_off = INST ? __hip_atomic_load(&inp->seed_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : itp->seed_off;
    const uint64_t item_trace_off = INST ? __hip_atomic_load(&inp->trace_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : itp->trace_off;
    const PgGraphDir gd = a.graphs[graph].dir[DIR];
    const uint32_t* __restrict__ smeta = a.colmeta + gd.meta_off;
    const PgNode* __restrict__ nodes = a.nodes + gd.node_off;
    const uint32_t n_nodes = gd.n_nodes;
    // (the item's seed region holds both halves' seeds: GROUPS * GL * C = 64 x the 16-lane variant's rows either way)
    uint32_t* nodekey = (uint32_t*)(a.workspace + item_seed_off + pg_seed_region_bytes(WIDE ? PG_VAR_WIDE + C * (GL / 16) : C, n_nodes));
    unsigned long long* nodekey64 = (unsigned long long*)nodekey;  // [n_nodes][4 reads][2 strands] 64-bit slots

    // ---- the moving frame ----------------------------------------------------------------------------------------------------
    // Quantities of pipeline step t are held as the f16 number 1024 + score + tau(t), tau(t) = PG_TAU0 + (t & 255): the frame
    // moves up by one per step.  The horizontal gap E decays by the gap-extension cost 1 per column = per step, so in the
    // moving frame it does not change at all: E' = max(E - 1, h - 6, 0) becomes max(E, h - 5, floor(t + 1)) -- the decrement is
    // gone (6 instead of 7 instructions per cell pair); the diagonal term pays for it with a constant (+1, folded into the
    // profile: diag + s + 1; +2 for a lane's first row, whose diagonal input is two steps old) and the floor of local
    // alignment with a wave-uniform constant per step (an SGPR).  The vertical gap F lives inside one step and keeps its
    // decrement.  Every 256 steps 256 is subtracted from all state, which keeps everything below 2048, where f16 holds
    // integers exactly (range used: 1024 - 300 .. 1024 + 512 + 8 + 256 + 2).  What leaves the registers is converted with
    // integer arithmetic on the bit patterns (0x6400 + n for 1024 + n): seeds and node maxima to plain scores on the rare
    // paths; the H trace keeps the low byte of score + tau, which pg_trace.hip undoes per cell.
    const uint32_t PADPK = pk_delta2(PAD + 1);  // (integer deltas on the bit patterns: pg_pk16.h)
    const uint32_t BIAS2 = PG_F16_BIAS2;  // the score 0 in frame 0

    // ---- query profiles of the 8 fills into LDS (gssw_qP_byte), shifted by the frame step ------------------------------
    // (read index, offset and length are uniform per read: loaded once, not per profile entry)
#pragma unroll
    for (int g = 0; g < GROUPS; ++g)
    {
        // half A / half B of the registers: the read's two strands, or (INST) two instances of their own
        uint32_t eA = INST ? inst_at(0, g) : itp->read[(int)half * GROUPS + g], eB = INST ? inst_at(1, g) : eA;
        const uint32_t ridxA = eA == PG_NONE ? PG_NONE : (eA & ~PG_INST_RC), ridxB = eB == PG_NONE ? PG_NONE : (eB & ~PG_INST_RC);
        const bool rcA = INST ? eA != PG_NONE && (eA & PG_INST_RC) != 0u : false, rcB = INST ? eB != PG_NONE && (eB & PG_INST_RC) != 0u : true;
        uint32_t offA = 0, LA = 0, offB = 0, LB = 0;
        if (ridxA != PG_NONE)
        {
            offA = a.base_off[ridxA];
            LA = a.base_off[ridxA + 1] - offA;
        }
        if (INST)
        {
            if (ridxB != PG_NONE)
            {
                offB = a.base_off[ridxB];
                LB = a.base_off[ridxB + 1] - offB;
            }
        }
        else
        {
            offB = offA;
            LB = LA;
        }
        for (int row = lane; row < ROWS; row += 64)
        {
            uint32_t cA = 5u, cB = 5u;  // 5 = padding row
            // DIR 0: strand A = toUpper(bases), strand B = reverseComplement(bases)
            // DIR 1: strand A = toUpper(reverse(bases)), strand B = reverseComplement(reverse(bases))
            //        (GraphAligner.cpp:315-337)
            if ((uint32_t)row < LA)
            {
                const uint32_t f = (uint8_t)a.bases[offA + row];
                const uint32_t r = (uint8_t)a.bases[offA + LA - 1 - row];
                const uint32_t ch = !rcA ? (DIR == 0 ? upper_c(f) : upper_c(r)) : (DIR == 0 ? comp_c(r) : comp_c(f));
                cA = nt_code(ch);
            }
            if ((uint32_t)row < LB)
            {
                const uint32_t f = (uint8_t)a.bases[offB + row];
                const uint32_t r = (uint8_t)a.bases[offB + LB - 1 - row];
                const uint32_t ch = !rcB ? (DIR == 0 ? upper_c(f) : upper_c(r)) : (DIR == 0 ? comp_c(r) : comp_c(f));
                cB = nt_code(ch);
            }
            const int shift = (row % C) == 0 ? 2 : 1;  // a lane's first row takes its diagonal input from two steps back
#pragma unroll
            for (uint32_t code = 0; code < 4; ++code)
            {
                const int sA = (cA == 5u ? PAD : sub_score(code, cA)) + shift;
                const int sB = (cB == 5u ? PAD : sub_score(code, cB)) + shift;
                prof[(g * 4 + code) * ROWS + row] = pk_delta(sA, sB);
            }
        }
    }
    // (device-scope stores / loads on the keys, but only a workgroup-scope fence: an agent-scope fence would write the
    // whole L2 back, trace stores included)
    // this wavefront's key slots only ([node][4 reads][2 strands] 64-bit slots: 4 * GROUPS dwords per node from its first read)
    for (uint32_t e = lane; e < n_nodes * 4 * GROUPS; e += 64)
        __hip_atomic_store(&nodekey[(e / (4 * GROUPS)) * 16 + half * 4 * GROUPS + e % (4 * GROUPS)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_block();
    __syncthreads();
    // rows of this lane that exist in its read (the others are padding rows)
    uint32_t real_rows = 0, real_rows_hi = 0;  // (the two halves' reads differ in an instance item)
    {
        const uint32_t eA = INST ? inst_at(0, grp) : itp->read[grp];
        const uint32_t ridx = eA == PG_NONE ? PG_NONE : (INST ? eA & ~PG_INST_RC : eA);
        const uint32_t Lg = ridx == PG_NONE ? 0u : a.base_off[ridx + 1] - a.base_off[ridx];
        real_rows = Lg > (uint32_t)(k * C) ? Lg - (uint32_t)(k * C) : 0u;
        real_rows_hi = real_rows;
        if (INST)
        {
            const uint32_t eB = inst_at(1, grp);
            const uint32_t rb = eB == PG_NONE ? PG_NONE : (eB & ~PG_INST_RC);
            const uint32_t Lb = rb == PG_NONE ? 0u : a.base_off[rb + 1] - a.base_off[rb];
            real_rows_hi = Lb > (uint32_t)(k * C) ? Lb - (uint32_t)(k * C) : 0u;
        }
    }

    const uint32_t nsteps = pg_fill_steps_lanes(gd.ncols, GL);  // even
    // [node][lane][SEED_DW] / [step / 2][TRACE_DW][lane][step & 1] (one store instruction = 512 contiguous bytes); wavefront `half` of a wide
    // item owns the second n_nodes * 64 * SEED_DW / nsteps * 64 * TRACE_DW dwords
    uint32_t* __restrict__ seed = (uint32_t*)(a.workspace + item_seed_off) + (size_t)half * n_nodes * 64 * SEED_DW;
    uint32_t* __restrict__ trace = (uint32_t*)(a.workspace + item_trace_off) + (size_t)half * nsteps * 64 * TRACE_DW;

    // H of the previous column lives in one of two register sets (HA / HB): a step reads one and writes the other, and the
    // step loop is unrolled twice with the roles swapped -- no register-to-register copies at the loop edge (the same for the
    // profile rows of the current / next column, sA / sB).  Initial values: score 0 in the frames of steps -1 / -2 / 0.
    constexpr uint32_t ONE2 = 0x00010001u;
    uint32_t HA[C], HB[C], E[C];
#pragma unroll
    for (int r = 0; r < C; ++r)
    {
        HA[r] = BIAS2 + (PG_TAU0 - 1) * ONE2;
        HB[r] = BIAS2 + (PG_TAU0 - 2) * ONE2;
        E[r] = BIAS2 + PG_TAU0 * ONE2;
    }
    // F is handed down as "F + 1" (the receiving row uses it as it is where every other row decrements first)
    uint32_t Fsend = BIAS2;
    // What the row's first lane sees as "the lane above": score 0 two steps ago for the diagonal (incremented with the frame
    // every step), anything not above score 0 for F.  row_shr:1 never writes lane 0 of a row, so that lane keeps these while
    // the other lanes receive their neighbour's values every step.
    uint32_t dHin = BIAS2 + (PG_TAU0 - 3) * ONE2, Fin = BIAS2;
    // Seed cache: seeds this lane stored for successors that are not its neighbours in the layout stay in registers (SEED_DW
    // dwords each: entry A the last one of an even node, entry B -- where the registers allow it -- that of an odd node; with
    // one entry, A takes every node).  A seed that has to be LOADED at a node's first column stops the wavefront for a memory
    // round trip (behind all of its outstanding trace stores: one counter, in order) at every one of the GL steps in which a
    // lane of the read reaches that column: a tenth of the whole fill on the left flank -> allele -> right flank graphs of
    // single events, whose right flank wants the left flank's seed (and a swap's second allele / right flank likewise).  The
    // wide variants with 32 lanes per read are held to 10-13 wavefronts per CU by their LDS profile, not by registers: they
    // can afford the entries too.
    // (byte variants: one dword per row, bytes H_A, H_B, Enext_A, Enext_B; wide variants: H and Enext dwords in arrays of their own)
    constexpr int CACHE_N = SEEDCACHE ? C : 1, CACHE_NE = SEEDCACHE && WIDE ? C : 1;
    constexpr int CACHE_NB = SEEDCACHE2 ? C : 1, CACHE_NEB = SEEDCACHE2 && WIDE ? C : 1;
    uint32_t cseed[CACHE_N], cseedE[CACHE_NE];
    uint32_t cnode = 0xFFFFFFFFu;
    uint32_t cseedB[CACHE_NB], cseedEB[CACHE_NEB];
    uint32_t cnodeB = 0xFFFFFFFFu;
    uint32_t M = BIAS2 + (PG_TAU0 - 1) * ONE2, FC = 0;  // node maximum (frame of the previous step) / step that first reached it
    uint32_t FR = 0;  // WIDE: smallest row (within the lane) holding the lane's maximum in column FC, per strand
    const uint32_t NEG5 = pk_delta2(-5);  // gap extend - gap open, per half; NEG1: the vertical gap's extension
    const uint32_t NEG1 = pk_delta2(-1);
    const uint32_t NEG256 = 0xDC00DC00u;  // (-256.0, -256.0)
    const uint32_t* profl = prof + lgrp * 4 * ROWS + k * C;
    const uint32_t trace_lane_off = (uint32_t)lane * 8u;  // (a lane stores the dwords of an even step and of the odd step behind it together)

    // Column meta words: the word of step t is the same for the whole wavefront, so it is read with SCALAR loads through
    // the constant address space (the graph tables are never written by a kernel), two steps ahead.  Scalar loads count
    // on lgkmcnt: the step loop never has to wait on vmcnt, i.e. on its own trace stores.  The array is padded with
    // PG_META_PAD idle words on the host.
    typedef const __attribute__((address_space(4))) uint32_t* const_u32_ptr;
    const const_u32_ptr cmeta = (const_u32_ptr)(uintptr_t)smeta;
    uint32_t mw[PG_META_AHEAD];  // the words of the next PG_META_AHEAD steps, in scalar registers
#pragma unroll
    for (int q = 0; q < PG_META_AHEAD; ++q)
        mw[q] = cmeta[1 + q];
    // software pipeline: `meta` and the current profile rows always belong to the step about to be computed
    uint32_t meta = group_shr1_keep<GL>(cmeta[0], PG_META_IDLE);
    // code 4 (N / idle column: score 0 on real rows) in the profile's shifted form
    auto code4_rows = [&](uint32_t (&rows)[C]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < C; ++r)
            if (INST)
                rows[r] = pk_delta((uint32_t)r < real_rows ? (r == 0 ? 2 : 1) : PAD + 1, (uint32_t)r < real_rows_hi ? (r == 0 ? 2 : 1) : PAD + 1);
            else
                rows[r] = (uint32_t)r < real_rows ? (r == 0 ? pk_delta2(2) : pk_delta2(1)) : PADPK;
    };
    uint32_t sA[C], sB[C];
    {
        const uint32_t code = PG_META_CODE(meta);
        const uint32_t* pr = profl + (code & 3u) * ROWS;
#pragma unroll
        for (int r = 0; r < C; r += 2)
        {
            const uint2 v = *(const uint2*)(pr + r);
            sA[r] = v.x;
            sA[r + 1] = v.y;
        }
        if (code >= 4u)
            code4_rows(sA);
#pragma unroll
        for (int r = 0; r < C; ++r)
            sB[r] = 0;
    }
    // the H trace of a PAIR of steps (even, odd) is [TRACE_DW][64 lanes][2 steps] dwords behind a wave-uniform base that advances
    // by two steps every two steps: the dwords of the even step wait in registers for those of the odd one and leave as one
    // global_store_dwordx2 -- half the store instructions.  At C = 10 that changes nothing (the step is bound by VALU issue:
    // 8.93 / 8.93 ms per 200 k reads), the variants with more rows per lane gain (C = 12, 180 bp reads: +4 %; C = 16, 250 bp:
    // +2.8 %; profiles/r06_trace_pairs_ab.jsonl, r06_trace_pairs_readlen_ab.jsonl).  The traceback reads as many lines as before
    // (12.2 KB per read, profiles/traffic_r06.json): its diagonal leaves a dword for (step - 2, dword - 1), never for the pair's other half.
    // the stores take the scalar-base form (SGPR pair + 32-bit lane offset + immediate), no per-step vector address arithmetic
    // (readfirstlane: the base is uniform by construction; this makes it so for the compiler, which must keep it in SGPRs;
    // the builtin returns a signed int: without the casts the low half would be sign-extended over the high one)
    uint64_t tbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)(uintptr_t)trace >> 32)) << 32)
        | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)trace);
    constexpr uint32_t TRACE_STEP_BYTES = 64u * 4u * (uint32_t)TRACE_DW;
    // (the immediate of a global store is 13 bits, signed: beyond eight dwords per step the base sits 4 096 bytes into the pair)
    constexpr int TRACE_BIAS = TRACE_DW > 8 ? 4096 : 0;
    tbase += (uint64_t)TRACE_BIAS;

    // ---- one column of the affine-gap recurrence for C rows x 2 strands (Hin: previous column, Hout: this column) -----------
    // floorE = the bit pattern of score 0 in the frame of step t + 1; tau = tau(t); tvec = (t | t << 16), all wave-uniform
    uint32_t tpack[DIR == 0 ? TRACE_DW : 1];  // the even step's trace dwords
    auto column = [&](uint32_t (&Hin)[C], uint32_t (&Hout)[C], const uint32_t (&sc)[C], uint32_t dH, uint32_t Fabove, uint32_t floorE,
                      uint32_t tau, uint32_t tvec, const bool ODD) __attribute__((always_inline)) {
        (void)tau;  // (ODD is a literal at both call sites: the inlined copies keep one side of the test each)
        uint32_t diag = dH;
        uint32_t F = Fabove;
#pragma unroll
        for (int r = 0; r < C; ++r)
        {
            // the three additions are 32-bit integer additions on the bit patterns (two cycles where a packed one takes four)
            const uint32_t f = r == 0 ? F : F + NEG1;                        // F of this row in this step's frame
            const uint32_t h = pk_max3h(diag + sc[r], E[r], f);              // max(H(i-1,j-1) + s, E, F); E >= 0 is the local-alignment floor
            diag = Hin[r];
            Hout[r] = h;
            const uint32_t tt = h + NEG5;                                    // (h - gap open) in the next step's / next row's terms
            E[r] = pk_max3h_s(E[r], tt, floorE);                             // no decrement: the frame moves instead
            F = pk_maxu(f, tt);                                              // "F + 1" of the next row
        }
        Fsend = F;

        // ---- running node maximum + first step reaching it (gssw.c:369-386) -----------------------
        // three-input maxima: C rows + the running maximum (moved into this step's frame) in (C + 1) / 2 instructions
        const uint32_t Mprev = M + ONE2;  // into this step's frame (integer + 1 on the bit pattern = + 1.0 above 1024)
        uint32_t cm[C + 1];
#pragma unroll
        for (int r = 0; r < C; ++r)
            cm[r] = Hout[r];
        cm[C] = Mprev;
        const uint32_t Mn = pk_max_all<C + 1>(cm);
        if (DIR == 0 || WIDE)
        {
            // mask = 0xFFFF in the halves whose maximum grew: (Mprev - Mn) is negative there as a 16-bit integer
            uint32_t grew;
            asm("v_pk_sub_u16 %0, %1, %2\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=v"(grew) : "v"(Mprev), "v"(Mn));
            if (WIDE)
            {
                // The row of the maximum matters in one case only: a node whose final maximum is 251..255 -- the reference's
                // alignsEndAtMultNodes reads word-mode matrices through a byte pointer and then only sees the first half of the
                // node's cells (the tail below).  A maximum passes through each of those five values at most once, so the row
                // search runs a handful of times per node instead of at every growth step.  (The traceback's start row is found
                // by the traceback kernel from the H trace, like in the byte variants.)
                // the test itself runs at every step, so it is kept to five packed instructions: score - 251 per strand by one
                // subtraction of the wave-uniform pattern of 251 in this step's frame, clamped at 5 (anything below the window
                // wraps around to a large unsigned value and is clamped too), in the window where the result is not 5
                const uint32_t inwin = pk_minu(pk_sub(Mn, BIAS2 + (tau + 251u) * ONE2), 5u * ONE2) ^ (5u * ONE2);
                const uint32_t need = inwin & grew;
                if (need)
                {
                    const uint32_t mnA = Mn & 0xFFFFu, mnB = Mn >> 16;  // bit patterns: 0x6400 + score + tau
                    const bool needA = (need & 0xFFFFu) != 0u, needB = (need >> 16) != 0u;
                    uint32_t frA = FR & 0xFFFFu, frB = FR >> 16;
#pragma unroll
                    for (int r = C - 1; r >= 0; --r)
                    {
                        if (needA && (Hout[r] & 0xFFFFu) == mnA)
                            frA = (uint32_t)r;
                        if (needB && (Hout[r] >> 16) == mnB)
                            frB = (uint32_t)r;
                    }
                    FR = frA | (frB << 16);
                }
            }
            asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(FC) : "v"(grew), "s"(tvec));
        }
        M = Mn;

        if (DIR == 0)
        {
            // one byte per cell: the low byte of the bit pattern = (score + tau) mod 256 (all of the score in the byte variants;
            // in the wide ones the traceback keeps exact scores by following differences, which are small between neighbours).
            // A dword holds DIAGONAL neighbours: the odd row r + 1 of this column and the even row r of the column before it
            // (Hin, in its own step's frame) -- the traceback walks diagonals, and every memory read it makes costs a whole
            // 128-byte line (profiles/r03_sector_probe.json), so two cells of its path per dword halve the lines it pulls from
            // under the next chunk's fill.  Where a node begins in this column Hin has been turned into the seed by
            // first_column: the even rows of a node's LAST column are therefore not in the trace; they are in the seed
            // region, which forward-graph fills store for every node (last_column).
#pragma unroll
            for (int r = 0; r < C; r += 2)
            {
                const uint32_t packed = __builtin_amdgcn_perm(Hout[r + 1], Hin[r], 0x06020400u);  // bytes A_r', A_r+1, B_r', B_r+1 (one v_perm)
                if (!ODD)
                {
                    tpack[DIR == 0 ? r / 2 : 0] = packed;
                    // (made HERE: left alone the compiler sinks this v_perm to the store in the next step, and then keeps a second copy
                    // of the odd rows of this column for it across the next step's first-column block -- five v_mov per pair of steps)
                    asm volatile("" : "+v"(tpack[DIR == 0 ? r / 2 : 0]));
                }
                else
                {
                    const uint64_t both = (uint64_t)tpack[DIR == 0 ? r / 2 : 0] | ((uint64_t)packed << 32);
                    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 nt" : : "v"(trace_lane_off), "v"(both), "s"(tbase), "n"((r / 2) * 512 - TRACE_BIAS) : "memory");
                }
            }
        }
    };
    // ---- node boundaries ---------------------------------------------------------------------------------------------------
    // first column: seed = lane-wise max over the predecessors (gssw_create_seed_byte); the predecessor that directly
    // precedes this node in the layout is still in Hin / E.  Seeds are stored as plain scores; tau = tau(t).
    auto first_column = [&](uint32_t (&Hin)[C], uint32_t meta_cur, uint32_t tau) __attribute__((always_inline)) {
        const uint32_t node = PG_META_NODE(meta_cur);
        const uint32_t hshift = (tau - 1u) * ONE2, eshift = tau * ONE2;  // previous column's H lives one frame back
        // (the maxima are taken in place: no second copy of a column in registers)
        if (!(meta_cur & PG_META_PRED_ADJ))
        {  // nothing of the node before it in the layout flows in: start from score 0
#pragma unroll
            for (int r = 0; r < C; ++r)
            {
                mov_into(Hin[r], BIAS2 + hshift);
                E[r] = BIAS2 + eshift;
            }
        }
        if (!(meta_cur & PG_META_PRED_MANY))
        {
            // predecessor summary in the meta word: no table loads
            if (meta_cur & PG_META_PRED_ONE)
            {
                const uint32_t pid = (meta_cur >> PG_META_PRED_SHIFT) & 0x7Fu;
                // byte variants: bytes (H_A, H_B, Enext_A, Enext_B) -> (0x6400 | H_A, 0x6400 | H_B) (the 0x64 bytes come from the
                // constant); wide variants: dwords (H_A | H_B << 16), (Enext_A | Enext_B << 16), each field = 0x6400 | score; then
                // into the frame with an integer addition on the bit patterns.  Each source has its own copy of this: the wait
                // for a LOADED seed must not sit behind the join where the cached ones would pay it too.
                if (SEEDCACHE && pid == cnode)
                {
#pragma unroll
                    for (int r = 0; r < C; ++r)
                    {
                        if constexpr (WIDE)
                        {
                            pk_maxu_into(Hin[r], pk_add(cseed[SEEDCACHE ? r : 0], hshift));
                            E[r] = pk_maxu(E[r], pk_add(cseedE[SEEDCACHE ? r : 0], eshift));
                        }
                        else
                        {
                            pk_maxu_into(Hin[r], pk_add(__builtin_amdgcn_perm(BIAS2, cseed[SEEDCACHE ? r : 0], 0x07010500u), hshift));
                            E[r] = pk_maxu(E[r], pk_add(__builtin_amdgcn_perm(BIAS2, cseed[SEEDCACHE ? r : 0], 0x07030502u), eshift));
                        }
                    }
                }
                else if (SEEDCACHE2 && pid == cnodeB)
                {
#pragma unroll
                    for (int r = 0; r < C; ++r)
                    {
                        if constexpr (WIDE)
                        {
                            pk_maxu_into(Hin[r], pk_add(cseedB[SEEDCACHE2 ? r : 0], hshift));
                            E[r] = pk_maxu(E[r], pk_add(cseedEB[SEEDCACHE2 ? r : 0], eshift));
                        }
                        else
                        {
                            pk_maxu_into(Hin[r], pk_add(__builtin_amdgcn_perm(BIAS2, cseedB[SEEDCACHE2 ? r : 0], 0x07010500u), hshift));
                            E[r] = pk_maxu(E[r], pk_add(__builtin_amdgcn_perm(BIAS2, cseedB[SEEDCACHE2 ? r : 0], 0x07030502u), eshift));
                        }
                    }
                }
                else
                {
                    const uint32_t* sp = seed + ((size_t)pid * 64 + lane) * SEED_DW;
#pragma unroll
                    for (int r = 0; r < C; ++r)
                    {
                        if constexpr (WIDE)
                        {
                            pk_maxu_into(Hin[r], pk_add(sp[2 * r], hshift));
                            E[r] = pk_maxu(E[r], pk_add(sp[2 * r + 1], eshift));
                        }
                        else
                        {
                            const uint32_t w = sp[r];
                            pk_maxu_into(Hin[r], pk_add(__builtin_amdgcn_perm(BIAS2, w, 0x07010500u), hshift));
                            E[r] = pk_maxu(E[r], pk_add(__builtin_amdgcn_perm(BIAS2, w, 0x07030502u), eshift));
                        }
                    }
                }
            }
        }
        else
        {
            const PgNode nd = nodes[node];
            for (uint32_t p = 0; p < nd.n_pred; ++p)
            {
                const uint32_t pid = a.preds[nd.pred_off + p];
                if (pid + 1 == node)
                    continue;  // the adjacent predecessor is what Hin / E hold already
                const uint32_t* sp = seed + ((size_t)pid * 64 + lane) * SEED_DW;
#pragma unroll
                for (int r = 0; r < C; ++r)
                {
                    if (WIDE)
                    {  // dwords: (H_A | H_B << 16), (Enext_A | Enext_B << 16), each 16-bit field = 0x6400 | score
                        pk_maxu_into(Hin[r], pk_add(sp[2 * r], hshift));
                        E[r] = pk_maxu(E[r], pk_add(sp[2 * r + 1], eshift));
                    }
                    else
                    {
                        const uint32_t w = sp[r];  // bytes: H_A, H_B, Enext_A, Enext_B
                        pk_maxu_into(Hin[r], pk_add(__builtin_amdgcn_perm(BIAS2, w, 0x07010500u), hshift));
                        E[r] = pk_maxu(E[r], pk_add(__builtin_amdgcn_perm(BIAS2, w, 0x07030502u), eshift));
                    }
                }
            }
        }
        M = BIAS2 + hshift;  // maximum so far: score 0, in the previous step's frame (the column moves it on)
        FC = 0;
        FR = 0;
    };
    // last column: the seed for the successors, and the node's maximum into its key.  Hout is in the frame of step t, E
    // (already the next column's) in that of step t + 1.
    auto last_column = [&](const uint32_t (&Hout)[C], uint32_t meta_cur, uint32_t tau) __attribute__((always_inline)) {
        const uint32_t node = PG_META_NODE(meta_cur);
        if ((meta_cur & PG_META_SAVE) || DIR == 0)  // forward graph: every node (the even rows of its last column, see column())
        {
            uint32_t* sp = seed + ((size_t)node * 64 + lane) * SEED_DW;
            const uint32_t hshift = tau * ONE2, eshift = (tau + 1u) * ONE2;
            // only a seed that a far successor will want goes into the cache (SAVE; a forward-graph node without one is stored for
            // the traceback alone)
            const bool keep = SEEDCACHE && (meta_cur & PG_META_SAVE) != 0u;
            const bool toB = keep && SEEDCACHE2 && (node & 1u) != 0u, toA = keep && !toB;
#pragma unroll
            for (int r = 0; r < C; ++r)
            {
                const uint32_t hs = pk_sub(Hout[r], hshift), es = pk_sub(E[r], eshift);  // 0x6400 | score
                if constexpr (WIDE)
                {
                    sp[2 * r] = hs;
                    sp[2 * r + 1] = es;
                    // (selects on the values, not branches around the stores: the arrays must stay in registers)
                    if constexpr (SEEDCACHE2)
                    {
                        cseedB[r] = toB ? hs : cseedB[r];
                        cseedEB[r] = toB ? es : cseedEB[r];
                    }
                    if constexpr (SEEDCACHE)
                    {
                        cseed[r] = toA ? hs : cseed[r];
                        cseedE[r] = toA ? es : cseedE[r];
                    }
                }
                else
                {
                    const uint32_t w = __builtin_amdgcn_perm(es, hs, 0x06040200u);
                    sp[r] = w;
                    if constexpr (SEEDCACHE2)
                        cseedB[r] = toB ? w : cseedB[r];
                    if constexpr (SEEDCACHE)
                        cseed[r] = toA ? w : cseed[r];
                }
            }
            cnodeB = toB ? node : cnodeB;
            cnode = toA ? node : cnode;
        }
        // key: max (12 bits) | inverted column (16 bits; a direction has <= 65519 columns) | inverted lane (4 bits)
        const uint32_t kinv = (uint32_t)(15 - k);
        const uint32_t mA = (M & 0x3FFu) - tau, mB = ((M >> 16) & 0x3FFu) - tau;  // the scores under 0x6400 + tau of their f16 patterns
        const uint32_t cA = ((FC & 0xFFFFu) - (uint32_t)k) & 0xFFFFu, cB = ((FC >> 16) - (uint32_t)k) & 0xFFFFu;  // step -> column
        if (WIDE)
        {
            // key: max | inverted column (16 bits) | inverted row (16 bits)
            const uint32_t rA = (uint32_t)(k * C) + (FR & 0xFFFFu), rB = (uint32_t)(k * C) + (FR >> 16);
            if (mA)
                atomicMax(&nodekey64[node * 8 + grp * 2 + 0],
                          ((unsigned long long)mA << 32) | ((unsigned long long)(0xFFFFu - cA) << 16) | (0xFFFFu - rA));
            if (mB)
                atomicMax(&nodekey64[node * 8 + grp * 2 + 1],
                          ((unsigned long long)mB << 32) | ((unsigned long long)(0xFFFFu - cB) << 16) | (0xFFFFu - rB));
        }
        else
        {
            if (mA)
                atomicMax(&nodekey[(node * 8 + grp * 2 + 0) * 2], (mA << 20) | ((0xFFFFu - cA) << 4) | kinv);
            if (mB)
                atomicMax(&nodekey[(node * 8 + grp * 2 + 1) * 2], (mB << 20) | ((0xFFFFu - cB) << 4) | kinv);
        }
    };

    // ---- one pipeline step: reads Hin (previous column) / sc (this column's profile rows), writes Hout / sn (next column's) ---
    auto step = [&](uint32_t (&Hin)[C], uint32_t (&Hout)[C], uint32_t (&sc)[C], uint32_t (&sn)[C], uint32_t t, const bool odd_step) __attribute__((always_inline)) {
        const uint32_t meta_cur = meta;
        const uint32_t tau = PG_TAU0 + (t & 255u);                 // wave-uniform: scalar registers
        const uint32_t floorE = BIAS2 + (tau + 1u) * ONE2;         // score 0 in the next step's frame
        const uint32_t tvec = (t & 0xFFFFu) * ONE2;
        // Hout still holds the column before the previous one: its last row is what the next lane needs as its diagonal
        // input one step later (lane k + 1 works one column behind lane k).  The row's first lane has no lane above: it keeps
        // its own register, which follows the frame by one per step.
        dHin = dHin + ONE2;
        dHin = group_shr1_keep<GL>(dHin, Hout[C - 1]);
        Fin = group_shr1_keep<GL>(Fin, Fsend);
        const uint32_t dH = dHin, F = Fin;
        // the next step's meta word; the profile rows of the next column (PREFETCH) or of this one
        meta = group_shr1_keep<GL>(mw[0], meta_cur);
#pragma unroll
        for (int q = 0; q + 1 < PG_META_AHEAD; ++q)
            mw[q] = mw[q + 1];
        mw[PG_META_AHEAD - 1] = cmeta[t + 1 + PG_META_AHEAD];
        const uint32_t meta_rows = PREFETCH ? meta : meta_cur;
        uint32_t (&rows)[C] = PREFETCH ? sn : sc;
        {
            const uint32_t* pr = profl + (PG_META_CODE(meta_rows) & 3u) * ROWS;
#pragma unroll
            for (int r = 0; r < C; r += 2)
            {
                const uint2 v = *(const uint2*)(pr + r);
                rows[r] = v.x;
                rows[r + 1] = v.y;
            }
        }
        // ONE test for everything that is rare: a node boundary in this column, or a column that carries code 4.  The column
        // itself is outside the branches: with the lanes of a read skewed by one column each, a node boundary keeps SOME lane
        // of the wavefront in a rare path for 16 steps in a row, and a column inside the branch would then be executed twice,
        // once per side.
        const bool rare = (int32_t)meta_cur < 0;  // PG_META_RARE: node boundary here, or code 4 here / in the next column
        // the lane mask of that test, in scalar registers: both rare blocks (before and after the column) hang off it by scalar
        // branches, so the common path holds ONE vector compare
        const unsigned long long rare_lanes = __ballot(rare);
        if (rare_lanes != 0ull && rare)
        {
            if (meta_rows & 4u)  // N in the graph, or the idle columns behind its end
                code4_rows(rows);
            if (meta_cur & PG_META_FIRST)
                first_column(Hin, meta_cur, tau);
        }
        column(Hin, Hout, sc, dH, F, floorE, tau, tvec, odd_step);
        if (rare_lanes != 0ull)  // wave-uniform
        {
            uint32_t mc = meta_cur;
            asm volatile("" : "+v"(mc));  // (keeps the compare below inside the branch: the compiler would hoist it onto the common path)
            if (mc >= (PG_META_RARE | PG_META_LAST_HI))  // a LAST column (the only words with bits 31 and 30 set): one compare
                last_column(Hout, meta_cur, tau);
        }
        if (odd_step)
            tbase += 2u * TRACE_STEP_BYTES;
    };

    for (uint32_t t = 0; t < nsteps; t += 2)
    {
        if ((t & 255u) == 0u && t != 0u)
        {
            // keep the frame small: 256 off everything that carries it (wave-uniform branch, once per 256 steps)
#pragma unroll
            for (int r = 0; r < C; ++r)
            {
                HA[r] = pk_addh_s(HA[r], NEG256);
                HB[r] = pk_addh_s(HB[r], NEG256);
                E[r] = pk_addh_s(E[r], NEG256);
            }
            Fsend = pk_addh_s(Fsend, NEG256);
            dHin = pk_addh_s(dHin, NEG256);
            M = pk_addh_s(M, NEG256);
        }
        if (PREFETCH)
        {
            step(HA, HB, sA, sB, t, false);
            step(HB, HA, sB, sA, t + 1, true);
        }
        else
        {
            step(HA, HB, sA, sA, t, false);
            step(HB, HA, sA, sA, t + 1, true);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the key atomics (and the trace stores) above are invisible to the compiler's own counters

    // ---- per fill: max_node (first node with the strictly largest score, gssw.c:4015-4018), multi flag, end column ------------
    // A wavefront keeps its place on the SIMD until this is done, and every load here is a round trip to memory under the
    // traffic of 1 000 other wavefronts (and of the previous chunk's traceback): the key loads of all nodes go out together,
    // lane k of a read taking nodes k, k + GL, ..., and nothing of the H trace is read back -- the row of the end cell is found by
    // the traceback kernel, which reads that column anyway (pg_trace.hip).
    {
        uint32_t lw[2] = { 0u, 0u };  // (score << 16 | 0xFFFF - node) of the lane's first node with its largest score; 0 = none
        uint32_t lcnt[2] = { 0u, 0u };  // nodes of this lane with that score
        unsigned long long lkey[2] = { 0ull, 0ull };
        for (uint32_t n = (uint32_t)k; n < n_nodes; n += (uint32_t)GL)
        {
#pragma unroll
            for (int st = 0; st < 2; ++st)
            {
                unsigned long long key;
                uint32_t m;
                if (WIDE)
                {
                    key = __hip_atomic_load(&nodekey64[n * 8 + grp * 2 + st], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    m = (uint32_t)(key >> 32);
                }
                else
                {
                    key = __hip_atomic_load(&nodekey[(n * 8 + grp * 2 + st) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    m = (uint32_t)key >> 20;
                }
                if (lcnt[st] == 0u || m > (lw[st] >> 16))
                {
                    lw[st] = (m << 16) | (0xFFFFu - n);
                    lkey[st] = key;
                    lcnt[st] = 1u;
                }
                else if (m == (lw[st] >> 16))
                    ++lcnt[st];
            }
        }
        // across the read's lanes: the largest (score, smallest node); how many nodes carry that score; the winner's key
        uint32_t gw[2], gcnt[2];
        unsigned long long gkey[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
        {
            uint32_t w = lw[st];
#pragma unroll
            for (int off = GL / 2; off >= 1; off >>= 1)
            {
                const uint32_t o = (uint32_t)__shfl_xor((int)w, off, GL);
                w = o > w ? o : w;
            }
            gw[st] = w;
            uint32_t c = (lcnt[st] != 0u && (lw[st] >> 16) == (w >> 16)) ? lcnt[st] : 0u;
            uint32_t klo = lw[st] == w ? (uint32_t)lkey[st] : 0u, khi = lw[st] == w ? (uint32_t)(lkey[st] >> 32) : 0u;
            if (w == 0u)
                klo = khi = 0u;  // no node at all (cannot happen: a graph has nodes)
#pragma unroll
            for (int off = GL / 2; off >= 1; off >>= 1)
            {
                c += (uint32_t)__shfl_xor((int)c, off, GL);
                klo |= (uint32_t)__shfl_xor((int)klo, off, GL);
                if (WIDE)
                    khi |= (uint32_t)__shfl_xor((int)khi, off, GL);
            }
            gcnt[st] = c;
            gkey[st] = ((unsigned long long)khi << 32) | klo;
        }
        if (k < 2)
        {
            const int strand = k;
            const uint32_t best = (strand ? gw[1] : gw[0]) >> 16;
            const uint32_t bestnode = 0xFFFFu - ((strand ? gw[1] : gw[0]) & 0xFFFFu);
            const unsigned long long bestkey = strand ? gkey[1] : gkey[0];
            uint32_t cnt = strand ? gcnt[1] : gcnt[0];
            if (WIDE && best >= 251u)
            {
                // gssw redid this fill in its 16-bit word mode (score + bias >= 255, gssw.c:380, 4100-4104), but
                // alignsEndAtMultNodes scans len * readLen BYTES of each node's matrix through a uint8_t*
                // (GraphAligner.cpp:180-187): a node only counts if the top score (<= 255) sits in the low byte of one of its
                // first ceil(len * readLen / 2) cells, in (reference position, read position) order.
                cnt = 0;
                if (best <= 255u)
                {
                    const uint32_t ridx = itp->read[grp];
                    const uint32_t L = ridx == PG_NONE ? 0u : a.base_off[ridx + 1] - a.base_off[ridx];
                    for (uint32_t n = 0; n < n_nodes; ++n)
                    {
                        const unsigned long long key = __hip_atomic_load(&nodekey64[n * 8 + grp * 2 + strand], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((uint32_t)(key >> 32) != best)
                            continue;
                        const uint32_t col = 0xFFFFu - (uint32_t)((key >> 16) & 0xFFFFu);
                        const uint32_t row = 0xFFFFu - (uint32_t)(key & 0xFFFFu);
                        const uint64_t cell = (uint64_t)(col - nodes[n].col_start) * L + row;
                        if (cell < ((uint64_t)nodes[n].len * L + 1) / 2)
                            ++cnt;
                    }
                }
            }
            PgFillSummary fs;
            fs.score = (int32_t)best;
            fs.max_node = (int32_t)bestnode;
            fs.ref_end = -1;  // node-local column and row of the end cell: the traceback kernel's (from end_col / read_end)
            fs.read_end = 0;
            fs.end_col = -1;
            fs.multi = cnt > 1 ? 1 : 0;
            fs.pad[0] = fs.pad[1] = 0;
            if (best > 0 && DIR == 0)
            {
                // the column of the first cell holding `best`, and the first row of the lane that holds it there
                if (WIDE)
                {
                    // (the key's row field = lane * C + a row < C that is only exact for maxima of 251..255: it orders the lanes)
                    fs.end_col = (int32_t)(0xFFFFu - (uint32_t)((bestkey >> 16) & 0xFFFFu));
                    fs.read_end = (int32_t)((0xFFFFu - (uint32_t)(bestkey & 0xFFFFu)) / (uint32_t)C * (uint32_t)C);
                }
                else
                {
                    fs.end_col = (int32_t)(0xFFFFu - (((uint32_t)bestkey >> 4) & 0xFFFFu));
                    fs.read_end = (int32_t)((15u - ((uint32_t)bestkey & 15u)) * (uint32_t)C);
                }
            }
            a.fillsum[((size_t)item_idx * PG_GROUPS + grp) * 2 + strand] = fs;
        }
    }
}

// One launch covers both graph directions: forward-graph workgroups (they store the H trace) and reversed-graph ones (scores
// + multi flags only, about a fifth cheaper).  Workgroups go to the eight XCDs of the chip round-robin by index, so "even =
// forward, odd = reversed" would give four XCDs all the forward-graph work and the other four all the reversed-graph work, and
// the launch would last as long as the heavier half.  Instead the direction changes every EIGHT workgroups: XCD x gets
// workgroups x, x + 8, x + 16, ... = pair p forward, pair p reversed, pair p + 8 forward, ... -- the same mix on every XCD.
// With AF_REVERSE_GRAPH off (both_dirs == 0) every workgroup is a forward-graph one.
template <int C, bool WIDE, int GL = PG_GROUP_LANES>
__global__ __launch_bounds__(64) void pg_fill_kernel(PgFillArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // GL = 32: two wavefronts (workgroups) per work item, unit = 2 * pair + half
    constexpr uint32_t HALVES = GL / PG_GROUP_LANES;
    const uint32_t n_units = a.n_pairs * HALVES;
    if (a.both_dirs)
    {
        const uint32_t j = blockIdx.x >> 3;
        const uint32_t unit = (j >> 1) * 8u + (blockIdx.x & 7u);
        if (unit >= n_units)
            return;  // the grid is rounded up to whole runs of eight
        if (j & 1u)
            pg_fill_body<C, 1, WIDE, GL>(a, unit / HALVES, lds, unit % HALVES);
        else
            pg_fill_body<C, 0, WIDE, GL>(a, unit / HALVES, lds, unit % HALVES);
    }
    else
        pg_fill_body<C, 0, WIDE, GL>(a, blockIdx.x / HALVES, lds, blockIdx.x % HALVES);
}

// The lean pass's two launches (byte variants): MODE 2 = the reversed-graph fill of every work-item pair (both strands of its four
// reads), MODE 3 = the forward-graph fill of the instance items the pick kernel made from their outcome.
template <int C, int MODE> __global__ __launch_bounds__(64) void pg_fill_lean_kernel(PgFillArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (MODE == 2)
        pg_fill_body<C, 1, false, PG_GROUP_LANES>(a, blockIdx.x, lds, 0);
    else
        pg_fill_body<C, 0, false, PG_GROUP_LANES, true>(a, blockIdx.x, lds, 0);
    // (one item per workgroup: a grid-stride loop over the items -- a bounded, persistent forward grid beside the next chunk's
    // reversed-graph fills was tried, profiles/r06_lean_streams_ab.jsonl -- costs 16 registers and with them the fourth wavefront per SIMD)
}

// The lean stage in ONE launch (pg_launch_fill_lean_fused): a wavefront takes TWO work-item pairs of one run -- eight reads -- through
// all of it: the reversed-graph fills of both pairs (both strands each), the pick (lanes 0..7, one read each), the forward-graph
// fill of the eight X strands as ONE instance item (which it writes into the leader pair's slot and sweeps at once).  The other
// strands a record still needs -- those the reversed-graph fills already ask for and those X's own forward fill has just made
// necessary (GraphAligner.cpp:340-356 by cases, pg_trace.hip) -- it queues as the instance item of the partner's slot for the chunk's
// second, small forward launch.  No dependency between large launches: wavefronts are in their reversed-graph and forward-graph
// sweeps at different times, so the trace stores spread over the launch by themselves.  A run's last pair without a partner runs both
// strands of its four reads forward (half 0 = X, half 1 = the other strand): the plain forward fill, nothing left open.
struct PgLeanFusedArgs
{
    const PgPlanSegment* segments;
    uint32_t n_segments;
    const uint32_t* group_count;
    PgInstItem* inst;  // [pair slot]: X item in the leader's slot, the other strands' item in its partner's
    uint32_t* yloc;    // per read, out: (slot << 3 | group << 1 | half) of the forward fill of its other strand (| PG_YLOC_PENDING: queued for
                       // the second forward launch), PG_NONE if none runs
    uint32_t* ucount;  // the chunk's list of reads for the traceback's second look, and its length (zeroed before the launch)
    uint32_t* ulist;
    uint32_t seg_begin, n_seg;  // the chunk's runs: segments[seg_begin .. seg_begin + n_seg)
};

template <int C> __global__ __launch_bounds__(64) void pg_fill_lean_fused_kernel(PgFillArgs a, PgLeanFusedArgs f)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // Every workgroup holds work.  The first ceil(n_pairs / 2): workgroup w looks at pair slots 2 w and 2 w + 1 and takes the couple one of
    // them LEADS (a couple = slots rs + 2 k, rs + 2 k + 1 of a run starting at rs, both holding reads) -- never both: consecutive slots
    // of a run have ranks of different parity, and a slot that ends a run leads no couple.  The last n_seg: workgroup k takes the lone
    // last pair of the chunk's k-th run if the run has an odd number of pairs.  (One workgroup per pair slot with the partners returning
    // at once left half of the chip idle -- workgroups are dealt to XCDs and CUs round-robin, the leaders sat on every other one: 59 ms
    // per million reads instead of 34.)
    const uint32_t n_couple_groups = (a.n_pairs + 1u) / 2u;
    const uint32_t pair0 = a.item_begin / 2u;  // the chunk's first pair slot in the batch's plan
    uint32_t p;       // the leader's pair slot
    bool partner;
    auto pairs_with_reads = [&](const PgPlanSegment& sgm) -> uint32_t {
        const uint32_t count = f.group_count[sgm.group];  // (a group's active reads come first)
        if (count <= 4u * sgm.first_pair)
            return 0u;
        const uint32_t n = (count - 4u * sgm.first_pair + 3u) / 4u;
        return n < sgm.n_pairs ? n : sgm.n_pairs;
    };
    if (blockIdx.x < n_couple_groups)
    {
        const uint32_t s0 = pair0 + 2u * blockIdx.x;
        uint32_t lo = 0, hi = f.n_segments;  // last segment whose pair_begin <= s0 (wave-uniform)
        while (hi - lo > 1)
        {
            const uint32_t mid = (lo + hi) >> 1;
            if (f.segments[mid].pair_begin <= s0)
                lo = mid;
            else
                hi = mid;
        }
        PgPlanSegment sg = f.segments[lo];
        uint32_t ne = pairs_with_reads(sg);
        uint32_t r = s0 - sg.pair_begin;
        p = s0;
        if ((r & 1u) || r + 1u >= ne)
        {
            // not s0: its neighbour, in this run or as the first pair of the next one
            p = s0 + 1u;
            if (2u * blockIdx.x + 1u >= a.n_pairs)
                return;
            if (p >= sg.pair_begin + sg.n_pairs)
            {
                if (lo + 1u >= f.n_segments)
                    return;
                sg = f.segments[lo + 1u];
                ne = pairs_with_reads(sg);
            }
            r = p - sg.pair_begin;
            if ((r & 1u) || r + 1u >= ne)
                return;
        }
        partner = true;
    }
    else
    {
        const uint32_t k = f.seg_begin + (blockIdx.x - n_couple_groups);
        const PgPlanSegment sg = f.segments[k];
        const uint32_t ne = pairs_with_reads(sg);
        if (!(ne & 1u))
            return;
        p = sg.pair_begin + ne - 1u;
        partner = false;
    }
    const uint32_t i = p - pair0;
    // ---- reversed-graph fills, both strands, of the couple's pairs
    for (uint32_t j = 0; j < (partner ? 2u : 1u); ++j)
    {
        if (a.both_dirs == 7u)
            break;  // (timing probe 7: the forward sweep only, on stale summaries)
        pg_fill_body<C, 1, false, PG_GROUP_LANES>(a, i + j, lds, 0);
        __syncthreads();
    }
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wavefront's own stores are in the L2 once vmcnt is 0, and what it reads back it reads with device-scope loads: an agent-scope fence would write the whole L2 back, every wavefront's trace stores included)
    // ---- the pick: lane l < 8 looks at read (pair l >> 2, group l & 3)
    const uint32_t lane = threadIdx.x;
    const uint32_t j = (lane >> 2) & 1u, g = lane & 3u;
    uint32_t ridx = PG_NONE;
    int X = 0, mXr = 0, mYr = 0, SX = 0;
    if (lane < 8u && (j == 0u || partner))
    {
        ridx = a.items[2 * (size_t)(p + j)].read[g];
        if (ridx != PG_NONE)
        {
            const PgFillSummary* fsR = a.fillsum + ((size_t)(2 * (p + j) + 1) * PG_GROUPS + g) * 2;
            const int SA = __hip_atomic_load(&fsR[0].score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int SB = __hip_atomic_load(&fsR[1].score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            X = SA >= SB ? 0 : 1;
            SX = X ? SB : SA;
            mXr = __hip_atomic_load(&fsR[X].multi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mYr = __hip_atomic_load(&fsR[1 - X].multi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    (void)SX;
    const PgWorkItem* fw0 = a.items + 2 * (size_t)p;
    PgInstItem* itX = f.inst + p;
    uint32_t yl = PG_NONE;
    if (lane == 0u)
    {
        itX->graph = fw0->graph;
        itX->pad = 0;
        itX->trace_off = fw0->trace_off;
        itX->seed_off = fw0->seed_off;
    }
    const uint32_t x_of_group = (uint32_t)__shfl((int)X, (int)g);  // X of read g of the leader pair (lanes 0..3 hold it), for every lane
    if (lane < 8u)
    {
        // X of (pair j, group g) -> half j; a lone pair's other strands -> half 1 (the plain forward fill of that pair)
        uint32_t e = ridx == PG_NONE ? PG_NONE : (ridx | (X ? PG_INST_RC : 0u));
        if (!partner && j == 1u)
        {
            // (lanes 4..7 of a lone pair: the other strand of read g of pair 0)
            const uint32_t r0 = a.items[2 * (size_t)p].read[g];
            e = r0 == PG_NONE ? PG_NONE : (r0 | (x_of_group ? 0u : PG_INST_RC));
        }
        itX->inst[j][g] = e;
        if (!partner && j == 0u && ridx != PG_NONE)
            yl = (p << 3) | (g << 1) | 1u;
    }
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.both_dirs == 6u)
        return;  // (timing probe PG_LEAN_FUSED_PROBE=6: the reversed-graph sweeps and the pick only)
    // ---- forward-graph fills of the X strands (and, for a lone pair, of the other strands beside them)
    pg_fill_body<C, 0, false, PG_GROUP_LANES, true>(a, i, lds, 0);
    __syncthreads();
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (partner)
    {
        // ---- which other strands does a record still need?  (X not unique, Y not multi on the reversed graph.)  Their forward fills
        // are queued as the instance item of the partner's slot -- this couple's own -- for the chunk's second, small forward launch, and
        // the reads are listed for the traceback's second look; the first look passes them by (yloc's PENDING bit).  A third sweep HERE
        // would be a third copy of the fill body in this kernel: 106 KB of code against an instruction cache of 64 KB, and every node
        // boundary's rare path a miss -- 59 ms per million reads instead of 34 (profiles/r06_lean_fused_ab.jsonl).
        bool need = false;
        if (lane < 8u && ridx != PG_NONE)
        {
            const int mXf = __hip_atomic_load(&a.fillsum[((size_t)(2 * p) * PG_GROUPS + g) * 2 + j].multi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            need = (mXf || mXr) && !mYr;
        }
        const unsigned long long needs = __ballot(need);
        const PgWorkItem* fw1 = a.items + 2 * (size_t)(p + 1u);
        PgInstItem* itY = f.inst + (p + 1u);
        if (lane == 0u)
        {
            itY->graph = fw1->graph;
            itY->pad = needs != 0ull ? 1u : 0u;  // (1: an item of the second forward launch)
            itY->trace_off = fw1->trace_off;
            itY->seed_off = fw1->seed_off;
        }
        if (lane < 8u)
        {
            // slots in the order of the lanes: (half 0, group 0) first -- an item is empty iff that entry is
            const uint32_t rank = (uint32_t)__popcll(needs & ((1ull << lane) - 1ull));
            const uint32_t n_need = (uint32_t)__popcll(needs);
            if (need)
            {
                itY->inst[rank >> 2][rank & 3u] = ridx | (X ? 0u : PG_INST_RC);
                yl = ((p + 1u) << 3) | ((rank & 3u) << 1) | (rank >> 2) | PG_YLOC_PENDING;
                f.ulist[atomicAdd(f.ucount, 1u)] = ((p + j) << 2) | g;
            }
            if (lane >= n_need)
                itY->inst[lane >> 2][lane & 3u] = PG_NONE;
        }
    }
    if (lane < 8u && ridx != PG_NONE)
        f.yloc[ridx] = yl;
}

template <int C> static hipError_t launch_lean_fused_c(PgFillArgs args, const PgLeanFusedArgs& f, uint32_t n_pairs, hipStream_t stream)
{
    void (*fn)(PgFillArgs, PgLeanFusedArgs) = pg_fill_lean_fused_kernel<C>;
    const size_t lds = (size_t)(64 * 4 * C) * sizeof(uint32_t);
    static const uint32_t probe = [] {
        const char* e = getenv("PG_LEAN_FUSED_PROBE");
        return e ? (uint32_t)atoi(e) : 5u;
    }();
    args.both_dirs = probe;
    args.n_pairs = n_pairs;
    hipLaunchKernelGGL(fn, dim3((n_pairs + 1u) / 2u + f.n_seg), dim3(64), lds, stream, args, f);
    return hipGetLastError();
}

hipError_t pg_launch_fill_lean_fused(int V, const PgFillArgs& args, const PgPlanSegment* segments, uint32_t n_segments, const uint32_t* group_count,
                                     PgInstItem* inst, uint32_t* yloc, uint32_t* ucount, uint32_t* ulist, uint32_t seg_begin, uint32_t n_seg,
                                     uint32_t n_pairs, hipStream_t stream)
{
    if (n_pairs == 0)
        return hipSuccess;
    PgLeanFusedArgs f{ segments, n_segments, group_count, inst, yloc, ucount, ulist, seg_begin, n_seg };
    switch (V)
    {
    case 2: return launch_lean_fused_c<2>(args, f, n_pairs, stream);
    case 4: return launch_lean_fused_c<4>(args, f, n_pairs, stream);
    case 6: return launch_lean_fused_c<6>(args, f, n_pairs, stream);
    case 8: return launch_lean_fused_c<8>(args, f, n_pairs, stream);
    case 10: return launch_lean_fused_c<10>(args, f, n_pairs, stream);
    case 12: return launch_lean_fused_c<12>(args, f, n_pairs, stream);
    case 14: return launch_lean_fused_c<14>(args, f, n_pairs, stream);
    case 16: return launch_lean_fused_c<16>(args, f, n_pairs, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int C> static hipError_t launch_lean_c(PgFillArgs args, uint32_t n_pairs, int mode, hipStream_t stream)
{
    void (*fn)(PgFillArgs) = mode == 2 ? pg_fill_lean_kernel<C, 2> : pg_fill_lean_kernel<C, 3>;  // (modes 3 and 4: the same kernel)
    const size_t lds = (size_t)(64 * 4 * C) * sizeof(uint32_t);
    args.both_dirs = (uint32_t)mode;
    args.n_pairs = n_pairs;
    const uint32_t grid = n_pairs;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, args);
    return hipGetLastError();
}

hipError_t pg_launch_fill_lean(int V, const PgFillArgs& args, uint32_t n_pairs, int mode, hipStream_t stream)
{
    if (n_pairs == 0)
        return hipSuccess;
    switch (V)
    {
    case 2: return launch_lean_c<2>(args, n_pairs, mode, stream);
    case 4: return launch_lean_c<4>(args, n_pairs, mode, stream);
    case 6: return launch_lean_c<6>(args, n_pairs, mode, stream);
    case 8: return launch_lean_c<8>(args, n_pairs, mode, stream);
    case 10: return launch_lean_c<10>(args, n_pairs, mode, stream);
    case 12: return launch_lean_c<12>(args, n_pairs, mode, stream);
    case 14: return launch_lean_c<14>(args, n_pairs, mode, stream);
    case 16: return launch_lean_c<16>(args, n_pairs, mode, stream);
    default: return hipErrorInvalidValue;
    }
}

template <int C, bool WIDE = false, int GL = PG_GROUP_LANES>
static hipError_t launch_c(PgFillArgs args, uint32_t n_pairs, bool revg, hipStream_t stream)
{
    void (*fn)(PgFillArgs) = pg_fill_kernel<C, WIDE, GL>;
    const size_t lds = (size_t)(64 * 4 * C) * sizeof(uint32_t);  // [64 / GL reads][4 codes][GL * C rows]
    if (lds > 48 * 1024)
    {
        hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    args.both_dirs = revg ? 1u : 0u;
    args.n_pairs = n_pairs;
    const uint32_t n_units = n_pairs * (uint32_t)(GL / PG_GROUP_LANES);
    hipLaunchKernelGGL(fn, dim3(revg ? (n_units + 7u) / 8u * 16u : n_units), dim3(64), lds, stream, args);
    return hipGetLastError();
}

// Launches the forward-graph fills of n_pairs item pairs and (revg) their reversed-graph fills.  V = the chunk's variant code;
// wide32: the wide variants with 32 lanes per read (two wavefronts per item, half the rows per lane).
hipError_t pg_launch_fill(int V, const PgFillArgs& args, uint32_t n_pairs, bool revg, bool wide32, hipStream_t stream)
{
    if (n_pairs == 0)
        return hipSuccess;
    if (pg_var_wide(V) && wide32)
    {
        switch (pg_var_c(V))
        {
        case 16: return launch_c<8, true, 32>(args, n_pairs, revg, stream);
        case 20: return launch_c<10, true, 32>(args, n_pairs, revg, stream);
        case 24: return launch_c<12, true, 32>(args, n_pairs, revg, stream);
        case 28: return launch_c<14, true, 32>(args, n_pairs, revg, stream);
        case 32: return launch_c<16, true, 32>(args, n_pairs, revg, stream);
        default: return hipErrorInvalidValue;
        }
    }
    switch (V)
    {
    case 2: return launch_c<2>(args, n_pairs, revg, stream);
    case 4: return launch_c<4>(args, n_pairs, revg, stream);
    case 6: return launch_c<6>(args, n_pairs, revg, stream);
    case 8: return launch_c<8>(args, n_pairs, revg, stream);
    case 10: return launch_c<10>(args, n_pairs, revg, stream);
    case 12: return launch_c<12>(args, n_pairs, revg, stream);
    case 14: return launch_c<14>(args, n_pairs, revg, stream);
    case 16: return launch_c<16>(args, n_pairs, revg, stream);
    case PG_VAR_WIDE + 16: return launch_c<16, true>(args, n_pairs, revg, stream);
    case PG_VAR_WIDE + 20: return launch_c<20, true>(args, n_pairs, revg, stream);
    case PG_VAR_WIDE + 24: return launch_c<24, true>(args, n_pairs, revg, stream);
    case PG_VAR_WIDE + 28: return launch_c<28, true>(args, n_pairs, revg, stream);
    case PG_VAR_WIDE + 32: return launch_c<32, true>(args, n_pairs, revg, stream);
    default: return hipErrorInvalidValue;
    }
}
