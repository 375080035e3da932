// pg_klib_packed.hip -- the klib (ksw) stage in the packed two-strand wave form.
//
// Replaces, like the general kernels of pg_klib.hip (used for paths shorter than a read),
//   common::KlibAlignment::update                   src/c++/lib/common/Klib.cpp:144-164
//   ksw_i16 / ksw_align(KSW_XSTART) / ksw_global    external/klib/ksw.c:223-321, 330-355, 457-531
// but with the DP arithmetic of the gssw fill kernel (pg_fill.hip): one 64-lane wavefront = 4 x 16 lanes, two alignments
// in the 16-bit halves of every VGPR, scores carried as half-precision numbers so that the three-input packed maximum
// applies, lanes skewed by one column each so that the whole state lives in registers.
//
//   pg_klib_local_kernel<C>   ksw_align's first pass for every (read, path, strand): one wavefront = 4 reads of one graph, the
//     two strands of each read in the two halves, one sweep per path of the graph.  Per alignment: score, END cell (first
//     column holding the maximum; in that column the first row in ksw's striped memory order).  The row is found without
//     a second look at the column: every lane keeps a snapshot of its rows at the step its own maximum last grew; the
//     lanes whose maximum is the global one and whose snapshot is of the END column hold exactly the cells ksw scans.
//   (pg_klib_select_kernel, pg_klib.hip: replays the candidate heap on the scores; only the candidates the pick can look at
//     -- those holding the read's best score -- go on; typically 1-2 of the 2 x paths candidates of a read)
//   pg_klib_finish_kernel<C>  the rest of KlibAlignment::update for those: ksw_align's reverse pass (START cell; it stops
//     at the first column reaching the score, ksw.c:347-352 with KSW_XSTOP) and ksw_global over the window with its
//     traceback.  Two candidates per 16 lanes (one per half): different reads and different targets, so the target code
//     is per half here (a window of codes in LDS) and a profile row is put together from two LDS rows.  ksw_global's
//     direction bits are not formed bit by bit: d = 2 iff F == H, d = 1 iff E == H (and F != H), "E extends" iff the new
//     E differs from H - gapoe (the same for F); each is one clamped packed subtraction, four of them are summed into one
//     byte per cell with packed FMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_klib.h"
#include "pg_pk16.h"

namespace
{
constexpr uint32_t ONE2 = 0x00010001u;
constexpr uint32_t BIAS2 = PG_F16_BIAS2;
constexpr uint32_t NEG5 = 0xC500C500u;    // (-5.0, -5.0)
constexpr uint32_t NEG6 = 0xC600C600u;    // (-6.0, -6.0)
constexpr uint32_t NEG256 = 0xDC00DC00u;  // (-256.0, -256.0)
constexpr uint32_t NEGINF2 = 0xFC00FC00u; // (-inf, -inf)
// padding rows (beyond the read) score so low that they stay at the local-alignment floor: below minus the longest read
constexpr int kpad(int C) { return C > 16 ? PG_PAD_SCORE_WIDE : PG_PAD_SCORE; }
// ksw scoring of KlibAlignerImpl (KlibAligner.cpp:134-142): match 1, mismatch -4, first gap base 5 + 1, every further one 1 --
// the numbers of the gssw stage, which is why its recurrence is reused as it is
static_assert(PG_GAP_OPEN == 6 && PG_GAP_EXT == 1, "klib's gapo + gape / gape");

typedef const __attribute__((address_space(4))) uint32_t* const_u32_ptr;

__device__ __forceinline__ int kscore(uint32_t t, uint32_t q) { return ((t | q) & 4u) ? 0 : (t == q ? 1 : -4); }

__device__ __forceinline__ uint32_t grp_max(uint32_t v)
{
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1)
    {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 16);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t grp_min(uint32_t v)
{
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1)
    {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 16);
        v = o < v ? o : v;
    }
    return v;
}

// ---- local affine sweep (ksw_i16's recurrence), the moving-frame form of pg_fill.hip -------------------------------------
// State of one lane: C rows x 2 halves.  Quantities of step t are held as the f16 number 1024 + score + PG_TAU0 + (t & 255).
template <int C> struct LocalSweep
{
    uint32_t HA[C], HB[C], E[C], snap[C];
    uint32_t Fsend, dHin, Fin, M, FC, Msnap;

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int r = 0; r < C; ++r)
        {
            HA[r] = BIAS2 + (PG_TAU0 - 1) * ONE2;
            HB[r] = BIAS2 + (PG_TAU0 - 2) * ONE2;
            E[r] = BIAS2 + PG_TAU0 * ONE2;
            snap[r] = 0;
        }
        Fsend = BIAS2;
        dHin = BIAS2 + (PG_TAU0 - 3) * ONE2;
        Fin = BIAS2;
        M = BIAS2 + (PG_TAU0 - 1) * ONE2;
        FC = 0;
        Msnap = 0;
    }
    // every 256 steps, before the step: 256 off everything that carries the frame
    __device__ __forceinline__ void normalize()
    {
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
    // one column: Hin = previous column, Hout = this one (still holding the column before the previous one: its last row is
    // the next lane's diagonal input), sc = this column's profile rows (shifted by the frame step, +2 on the lane's first row)
    __device__ __forceinline__ void column(uint32_t (&Hin)[C], uint32_t (&Hout)[C], const uint32_t (&sc)[C], uint32_t t)
    {
        const uint32_t tau = PG_TAU0 + (t & 255u);
        const uint32_t floorE = BIAS2 + (tau + 1u) * ONE2;  // score 0 in the next step's frame
        const uint32_t tvec = (t & 0xFFFFu) * ONE2;
        dHin = pk_add(dHin, ONE2);
        dHin = row_shr1_keep(dHin, Hout[C - 1]);
        Fin = row_shr1_keep(Fin, Fsend);
        uint32_t diag = dHin, F = Fin;
#pragma unroll
        for (int r = 0; r < C; ++r)
        {
            const uint32_t f = r == 0 ? F : pk_dech(F);
            const uint32_t h = pk_max3h(pk_addh(diag, sc[r]), E[r], f);
            diag = Hin[r];
            Hout[r] = h;
            const uint32_t tt = pk_addh_s(h, NEG5);
            E[r] = pk_max3h_s(E[r], tt, floorE);
            F = pk_maxu(f, tt);
        }
        Fsend = F;
        // running maximum, the step it last grew at, and the lane's rows at that step
        const uint32_t Mprev = pk_add(M, ONE2);
        uint32_t cm[C + 1];
#pragma unroll
        for (int r = 0; r < C; ++r)
            cm[r] = Hout[r];
        cm[C] = Mprev;
        const uint32_t Mn = pk_max_all<C + 1>(cm);
        uint32_t grew;
        asm("v_pk_sub_u16 %0, %1, %2\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=v"(grew) : "v"(Mprev), "v"(Mn));
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(FC) : "v"(grew), "s"(tvec));
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(Msnap) : "v"(grew), "v"(Mn));
#pragma unroll
        for (int r = 0; r < C; ++r)
            asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(snap[r]) : "v"(grew), "v"(Hout[r]));
        M = Mn;
    }
    // Result of one half (all 16 lanes of the group return the same): the maximum, the first column holding it, and in that
    // column the row that comes first in ksw_i16's striped memory order (ksw.c:303-308: slen = (rows + 7) / 8 vectors of 8).
    __device__ __forceinline__ void result(int half, int k, uint32_t slen, int& score, int& col, int& row) const
    {
        const uint32_t pat = (Msnap >> (16 * half)) & 0xFFFFu;
        const uint32_t fc = (FC >> (16 * half)) & 0xFFFFu;
        uint32_t key = 0;
        if (pat)
        {
            const uint32_t sc = (pat & 0x3FFu) - (PG_TAU0 + (fc & 255u));
            const uint32_t c = (fc - (uint32_t)k) & 0xFFFFu;
            key = sc ? ((sc << 16) | (0xFFFFu - c)) : 0u;
        }
        const uint32_t best = grp_max(key);
        uint32_t rk = 0x7FFFFFFFu;
        if (best != 0u && key == best)
        {
#pragma unroll
            for (int r = 0; r < C; ++r)
                if (((snap[r] >> (16 * half)) & 0xFFFFu) == pat)
                {
                    const uint32_t rw = (uint32_t)(k * C + r);
                    const uint32_t kk = (rw % slen) * 8u + rw / slen;
                    rk = kk < rk ? kk : rk;
                }
        }
        rk = grp_min(rk);
        if (best == 0u)
        {
            score = 0;
            col = -1;
            row = 0;
            return;
        }
        score = (int)(best >> 16);
        col = (int)(0xFFFFu - (best & 0xFFFFu));
        row = (int)(rk / 8u + (rk % 8u) * slen);
    }
};

// bases of one read strand as ksw codes (Klib.cpp:144-153: the reverse strand is reverseComplement() of the raw bases)
struct Strand
{
    const char* bases;
    int L;
    bool reverse;
    __device__ __forceinline__ uint32_t code(int j) const
    {
        return reverse ? (uint32_t)klib_code(klib_comp_raw((uint8_t)bases[L - 1 - j])) : (uint32_t)klib_code((uint8_t)bases[j]);
    }
};

// =================================================================================================================
// first pass: score + END cell of every (read, path, strand)
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(64) void pg_klib_local_kernel(KlibArgs a)
{
    constexpr int ROWS = PG_GROUP_LANES * C;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* prof = lds;  // [4 reads][4 codes][ROWS] packed (forward strand | reverse strand << 16)
    const int lane = threadIdx.x;
    const int grp = lane >> 4;
    const int k = lane & 15;

    const PgWorkItem* itp = a.work + 2 * (size_t)(a.pair_begin + blockIdx.x);
    if (itp->read[0] == PG_NONE)
        return;  // an empty slot of a plan re-written by the cascade's hand-over (active reads come first in a group)
    const LGraphDev g = a.graphs[itp->graph];
    constexpr int KPAD = kpad(C);
    const uint32_t PADPK = f16_bits(KPAD + 1) | (f16_bits(KPAD + 1) << 16);

#pragma unroll
    for (int gi = 0; gi < PG_GROUPS; ++gi)
    {
        const uint32_t ridx = itp->read[gi];
        uint32_t off = 0, L = 0;
        if (ridx != PG_NONE)
        {
            off = a.base_off[ridx];
            L = a.base_off[ridx + 1] - off;
        }
        for (int row = lane; row < ROWS; row += 64)
        {
            uint32_t cA = 5u, cB = 5u;  // 5 = padding row
            if ((uint32_t)row < L)
            {
                cA = (uint32_t)klib_code((uint8_t)a.bases[off + row]);
                cB = (uint32_t)klib_code(klib_comp_raw((uint8_t)a.bases[off + L - 1 - row]));
            }
            const int shift = (row % C) == 0 ? 2 : 1;
#pragma unroll
            for (uint32_t code = 0; code < 4; ++code)
            {
                const int sA = (cA == 5u ? KPAD : kscore(code, cA)) + shift;
                const int sB = (cB == 5u ? KPAD : kscore(code, cB)) + shift;
                prof[(gi * 4 + code) * ROWS + row] = f16_bits(sA) | (f16_bits(sB) << 16);
            }
        }
    }
    __syncthreads();
    const uint32_t ridx = itp->read[grp];
    const uint32_t Lg = ridx == PG_NONE ? 0u : a.base_off[ridx + 1] - a.base_off[ridx];
    const uint32_t real_rows = Lg > (uint32_t)(k * C) ? Lg - (uint32_t)(k * C) : 0u;
    const uint32_t slen = (Lg + 7u) / 8u;
    const uint32_t* profl = prof + grp * 4 * ROWS + k * C;
    const uint32_t per_read = 2u * a.max_paths;

    auto code4_rows = [&](uint32_t (&rows)[C]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < C; ++r)
            rows[r] = (uint32_t)r < real_rows ? (r == 0 ? 0x40004000u : 0x3C003C00u) : PADPK;  // score 0: (2.0, 2.0) / (1.0, 1.0)
    };

    for (uint32_t pi = 0; pi < g.n_paths; ++pi)
    {
        const LPathDev p = a.paths[g.path_off + pi];
        const const_u32_ptr cmeta = (const_u32_ptr)(uintptr_t)(a.pathmeta + p.meta_off);
        const uint32_t nsteps = pg_fill_steps(p.len);
        LocalSweep<C> S;
        S.init();
        // column codes: the word of step t is wave-uniform (scalar loads, two steps ahead) and rides down the lanes of a read
        // with one DPP move per step; the profile rows of the NEXT column are fetched while this one is computed
        uint32_t mw1 = cmeta[1], mw2 = cmeta[2];
        uint32_t meta = row_shr1_keep(cmeta[0], PG_META_IDLE);
        uint32_t sA[C], sB[C];
        {
            const uint32_t* pr = profl + (meta & 3u) * ROWS;
#pragma unroll
            for (int r = 0; r < C; r += 2)
            {
                const uint2 v = *(const uint2*)(pr + r);
                sA[r] = v.x;
                sA[r + 1] = v.y;
            }
            if (meta & 4u)
                code4_rows(sA);
#pragma unroll
            for (int r = 0; r < C; ++r)
                sB[r] = 0;
        }
        auto step = [&](uint32_t (&Hin)[C], uint32_t (&Hout)[C], uint32_t (&sc)[C], uint32_t (&sn)[C], uint32_t t) __attribute__((always_inline)) {
            const uint32_t meta_cur = meta;
            meta = row_shr1_keep(mw1, meta_cur);
            mw1 = mw2;
            mw2 = cmeta[t + 3];
            {
                const uint32_t* pr = profl + (meta & 3u) * ROWS;
#pragma unroll
                for (int r = 0; r < C; r += 2)
                {
                    const uint2 v = *(const uint2*)(pr + r);
                    sn[r] = v.x;
                    sn[r + 1] = v.y;
                }
            }
            if (meta & 4u)  // N on the path, or the idle columns behind its end
                code4_rows(sn);
            S.column(Hin, Hout, sc, t);
        };
        for (uint32_t t = 0; t < nsteps; t += 2)
        {
            if ((t & 255u) == 0u && t != 0u)
                S.normalize();
            step(S.HA, S.HB, sA, sB, t);
            step(S.HB, S.HA, sB, sA, t + 1);
        }
        if (ridx != PG_NONE)
        {
#pragma unroll
            for (int half = 0; half < 2; ++half)
            {
                int score, te, qe;
                S.result(half, k, slen, score, te, qe);
                if (k == half)
                {
                    KlibItem out{};
                    if (score > 0)
                    {
                        out.score = score;
                        out.te = te;
                        out.qe = qe;
                        out.valid = 1u;
                    }
                    a.items[(size_t)ridx * per_read + 2u * pi + (uint32_t)half] = out;
                }
            }
        }
    }
}

// =================================================================================================================
// second pass + global alignment of the selected candidates
// =================================================================================================================
struct FinishInfo
{  // one candidate (LDS)
    uint32_t item;      // index into items[]; PG_NONE = empty slot
    uint32_t off;       // read bases
    int32_t L;
    uint32_t reverse;
    uint32_t seq_off;   // path codes
    int32_t score, te, qe, tb, qb;
    uint32_t pad[2];
};

constexpr int finish_win(int C) { return 2 * PG_GROUP_LANES * C + 16; }   // columns of target staged per candidate
constexpr int finish_winp(int C) { return 16 + finish_win(C) + 48; }      // + idle head (negative columns) and tail

template <int C>
__global__ __launch_bounds__(64) void pg_klib_finish_kernel(KlibArgs a)
{
    constexpr int ROWS = PG_GROUP_LANES * C;
    constexpr int WIN = finish_win(C);
    constexpr int WINP = finish_winp(C);
    constexpr int ZDW = C / 2;  // dwords of direction bytes per lane per step
    constexpr int KPAD = kpad(C);
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* prof = lds;                                        // [4 groups][5 codes][ROWS] packed (candidate A | candidate B << 16)
    uint16_t* win = (uint16_t*)(prof + PG_GROUPS * 5 * ROWS);    // [4 groups][WINP] (code A | code B << 8)
    FinishInfo* info = (FinishInfo*)(win + PG_GROUPS * WINP);    // [8]
    const int lane = threadIdx.x;
    const int grp = lane >> 4;
    const int k = lane & 15;
    const uint32_t per_read = 2u * a.max_paths;
    uint32_t* __restrict__ zw = (uint32_t*)(a.z + (size_t)blockIdx.x * a.z_bytes);  // [step][ZDW][64 lanes]

    // (the number of selected candidates is read where the select kernel left it: the host does not wait for it to size this launch)
    const uint32_t n_work = a.work_count ? *a.work_count : a.n_work;
    for (uint32_t base = blockIdx.x * 8u; base < n_work; base += gridDim.x * 8u)
    {
        __syncthreads();
        if (lane < 8)
        {
            FinishInfo fi{};
            fi.item = PG_NONE;
            fi.te = -1;
            fi.qe = -1;
            const uint32_t w = base + (uint32_t)lane;
            if (w < n_work)
            {
                const uint32_t item = a.worklist[w];
                const uint32_t r = item / per_read, sub = item % per_read;
                const KlibItem ki = a.items[item];
                const LGraphDev g = a.graphs[a.graph_of_read[r]];
                const LPathDev p = a.paths[g.path_off + (sub >> 1)];
                fi.item = item;
                fi.off = a.base_off[r];
                fi.L = (int32_t)(a.base_off[r + 1] - fi.off);
                fi.reverse = sub & 1u;
                fi.seq_off = p.seq_off;
                fi.score = ki.score;
                fi.te = ki.te;
                fi.qe = ki.qe;
            }
            info[lane] = fi;
        }
        __syncthreads();

        // ---------------------------------------------------------------------------------------------------------
        // reverse pass: rows = query[qe .. 0], columns = target[te .. 0]; stops once every candidate of the wavefront has
        // reached its score (ksw.c:347-352)
        // ---------------------------------------------------------------------------------------------------------
        uint32_t max_cols = 0;
#pragma unroll
        for (int gi = 0; gi < PG_GROUPS; ++gi)
        {
            const FinishInfo fa = info[2 * gi], fb = info[2 * gi + 1];
            const Strand qa{ a.bases + fa.off, fa.L, fa.reverse != 0 }, qb{ a.bases + fb.off, fb.L, fb.reverse != 0 };
            for (int row = lane; row < ROWS; row += 64)
            {
                const uint32_t cA = row <= fa.qe ? qa.code(fa.qe - row) : 5u;
                const uint32_t cB = row <= fb.qe ? qb.code(fb.qe - row) : 5u;
                const int shift = (row % C) == 0 ? 2 : 1;
#pragma unroll
                for (uint32_t code = 0; code < 5; ++code)
                {
                    const int sA = (cA == 5u ? KPAD : kscore(code, cA)) + shift;
                    const int sB = (cB == 5u ? KPAD : kscore(code, cB)) + shift;
                    prof[(gi * 5 + code) * ROWS + row] = f16_bits(sA) | (f16_bits(sB) << 16);
                }
            }
            for (int col = lane; col < WINP; col += 64)
            {
                const int idx = col - 16;
                const uint32_t cl = (idx >= 0 && idx <= fa.te) ? a.pathcode[fa.seq_off + (uint32_t)(fa.te - idx)] : 4u;
                const uint32_t ch = (idx >= 0 && idx <= fb.te) ? a.pathcode[fb.seq_off + (uint32_t)(fb.te - idx)] : 4u;
                win[gi * WINP + col] = (uint16_t)(cl | (ch << 8));
            }
            const uint32_t ca = (uint32_t)(fa.te + 1), cb = (uint32_t)(fb.te + 1);
            max_cols = max_cols > ca ? max_cols : ca;
            max_cols = max_cols > cb ? max_cols : cb;
        }
        __syncthreads();
        const FinishInfo fA = info[2 * grp], fB = info[2 * grp + 1];
        const uint32_t* profl = prof + grp * 5 * ROWS + k * C;
        const uint16_t* winl = win + grp * WINP + (16 - k);  // winl[t] = codes of column t - k
        int rsc[2], rte[2], rqe[2];
        {
            max_cols = max_cols < (uint32_t)WIN ? max_cols : (uint32_t)WIN;
            const uint32_t nsteps = pg_fill_steps(max_cols);
            LocalSweep<C> S;
            S.init();
            uint32_t w1 = winl[1];
            uint32_t sA[C], sB[C];
            auto fetch = [&](uint32_t wcode, uint32_t (&rows)[C]) __attribute__((always_inline)) {
                const uint32_t* plo = profl + (wcode & 7u) * ROWS;
                const uint32_t* phi = profl + ((wcode >> 8) & 7u) * ROWS;
#pragma unroll
                for (int r = 0; r < C; r += 2)
                {
                    const uint2 lo = *(const uint2*)(plo + r);
                    const uint2 hi = *(const uint2*)(phi + r);
                    rows[r] = (lo.x & 0xFFFFu) | (hi.x & 0xFFFF0000u);
                    rows[r + 1] = (lo.y & 0xFFFFu) | (hi.y & 0xFFFF0000u);
                }
            };
            fetch(winl[0], sA);
#pragma unroll
            for (int r = 0; r < C; ++r)
                sB[r] = 0;
            auto step = [&](uint32_t (&Hin)[C], uint32_t (&Hout)[C], uint32_t (&sc)[C], uint32_t (&sn)[C], uint32_t t) __attribute__((always_inline)) {
                const uint32_t wn = w1;
                w1 = winl[t + 2];
                fetch(wn, sn);
                S.column(Hin, Hout, sc, t);
            };
            uint32_t stop_at = 0xFFFFFFFFu;
            for (uint32_t t = 0; t < nsteps && t < stop_at; t += 2)
            {
                if ((t & 15u) == 0u && t != 0u && stop_at == 0xFFFFFFFFu)
                {
                    // M is in the frame of step t - 1 here (before the normalisation below)
                    const uint32_t tau = PG_TAU0 + ((t - 1u) & 255u);
                    const uint32_t tgtA = 0x6400u + (uint32_t)fA.score + tau, tgtB = 0x6400u + (uint32_t)fB.score + tau;
                    const unsigned long long ba = __ballot((S.M & 0xFFFFu) >= tgtA), bb = __ballot((S.M >> 16) >= tgtB);
                    bool all = true;
#pragma unroll
                    for (int gi = 0; gi < PG_GROUPS; ++gi)
                        all = all && ((ba >> (16 * gi)) & 0xFFFFull) != 0ull && ((bb >> (16 * gi)) & 0xFFFFull) != 0ull;
                    if (all)
                        stop_at = t + 16u;  // the lanes behind still have to pass the columns before the one that reached it
                }
                if ((t & 255u) == 0u && t != 0u)
                    S.normalize();
                step(S.HA, S.HB, sA, sB, t);
                step(S.HB, S.HA, sB, sA, t + 1);
            }
            S.result(0, k, (uint32_t)(fA.qe + 1 + 7) / 8u, rsc[0], rte[0], rqe[0]);
            S.result(1, k, (uint32_t)(fB.qe + 1 + 7) / 8u, rsc[1], rte[1], rqe[1]);
        }
        if (k < 2)
        {
            const FinishInfo& f = k == 0 ? fA : fB;
            if (f.item != PG_NONE)
            {
                const int tb = f.te - rte[k], qb = f.qe - rqe[k];
                if (rsc[k] != f.score || tb < 0 || qb < 0 || f.te - tb + 1 > WIN)
                {  // cannot happen (the reverse pass reaches the score of the first one inside 2 x read length columns)
                    atomicOr(a.error, 4u);
                    info[2 * grp + k].item = PG_NONE;
                    info[2 * grp + k].te = -1;
                    info[2 * grp + k].qe = -1;
                    info[2 * grp + k].tb = 0;
                    info[2 * grp + k].qb = 0;
                    a.items[f.item].valid = 0u;
                }
                else
                {
                    info[2 * grp + k].tb = tb;
                    info[2 * grp + k].qb = qb;
                }
            }
        }
        __syncthreads();

        // ---------------------------------------------------------------------------------------------------------
        // ksw_global over the window (rows query[qb..qe], columns target[tb..te]; band = path length >= read length: the
        // whole rectangle), one direction byte per cell: bit0 E != H, bit1 F != H, bit2 E extends, bit3 F extends
        // ---------------------------------------------------------------------------------------------------------
        uint32_t max_tl = 0;
#pragma unroll
        for (int gi = 0; gi < PG_GROUPS; ++gi)
        {
            const FinishInfo fa = info[2 * gi], fb = info[2 * gi + 1];
            const Strand qa{ a.bases + fa.off, fa.L, fa.reverse != 0 }, qb{ a.bases + fb.off, fb.L, fb.reverse != 0 };
            const int qla = fa.qe - fa.qb + 1, qlb = fb.qe - fb.qb + 1;  // 0 for an empty slot (qe = -1, qb = 0)
            const int tla = fa.te - fa.tb + 1, tlb = fb.te - fb.tb + 1;
            for (int row = lane; row < ROWS; row += 64)
            {
                const uint32_t cA = row < qla ? qa.code(fa.qb + row) : 4u;
                const uint32_t cB = row < qlb ? qb.code(fb.qb + row) : 4u;
#pragma unroll
                for (uint32_t code = 0; code < 5; ++code)
                    prof[(gi * 5 + code) * ROWS + row] = f16_bits(kscore(code, cA)) | (f16_bits(kscore(code, cB)) << 16);
            }
            for (int col = lane; col < WINP; col += 64)
            {
                const int idx = col - 16;
                const uint32_t cl = (idx >= 0 && idx < tla) ? a.pathcode[fa.seq_off + (uint32_t)(fa.tb + idx)] : 4u;
                const uint32_t ch = (idx >= 0 && idx < tlb) ? a.pathcode[fb.seq_off + (uint32_t)(fb.tb + idx)] : 4u;
                win[gi * WINP + col] = (uint16_t)(cl | (ch << 8));
            }
            max_tl = max_tl > (uint32_t)tla ? max_tl : (uint32_t)tla;
            max_tl = max_tl > (uint32_t)tlb ? max_tl : (uint32_t)tlb;
        }
        __syncthreads();
        const FinishInfo gA = info[2 * grp], gB = info[2 * grp + 1];
        {
            const int tla = gA.te - gA.tb + 1, tlb = gB.te - gB.tb + 1;
            const uint32_t ncols_g = (uint32_t)(tla > tlb ? tla : tlb);
            const uint32_t nsteps = pg_fill_steps(max_tl);
            uint32_t HA[C], HB[C], E[C];
#pragma unroll
            for (int r = 0; r < C; ++r)
            {
                const uint32_t hb = f16_bits(-(6 + k * C + r));  // H(-1, j) = -(gapo + gape * (j + 1)) (ksw.c:475-478)
                HA[r] = HB[r] = hb | (hb << 16);
                E[r] = NEGINF2;
            }
            uint32_t X = 0;  // lane 0 of a read: H(i - 1, -1) of the column about to be computed: 0, then -(gapo + gape * i)
            uint32_t Fsend = NEGINF2, dHin = 0, Fin = NEGINF2;
            const uint32_t B1024 = BIAS2, EIGHT2 = 0x48004800u;
            uint32_t w1 = winl[1];
            uint32_t sA[C], sB[C];
            auto fetch = [&](uint32_t wcode, uint32_t (&rows)[C]) __attribute__((always_inline)) {
                const uint32_t* plo = profl + (wcode & 7u) * ROWS;
                const uint32_t* phi = profl + ((wcode >> 8) & 7u) * ROWS;
#pragma unroll
                for (int r = 0; r < C; r += 2)
                {
                    const uint2 lo = *(const uint2*)(plo + r);
                    const uint2 hi = *(const uint2*)(phi + r);
                    rows[r] = (lo.x & 0xFFFFu) | (hi.x & 0xFFFF0000u);
                    rows[r + 1] = (lo.y & 0xFFFFu) | (hi.y & 0xFFFF0000u);
                }
            };
            fetch(winl[0], sA);
#pragma unroll
            for (int r = 0; r < C; ++r)
                sB[r] = 0;
            auto step = [&](uint32_t (&Hin)[C], uint32_t (&Hout)[C], uint32_t (&sc)[C], uint32_t (&sn)[C], uint32_t t) __attribute__((always_inline)) {
                const uint32_t wn = w1;
                w1 = winl[t + 2];
                fetch(wn, sn);
                // the lane above: its last row of the column before the previous one (diagonal input), its running F of this column
                dHin = row_shr1_keep(X, Hout[C - 1]);
                Fin = row_shr1_keep(Fin, Fsend);
                X = t == 0u ? NEG6 : pk_dech(X);
                const uint32_t i = t - (uint32_t)k;
                if (t >= (uint32_t)k && i < ncols_g)
                {
                    uint32_t diag = dHin, f = Fin;
                    uint32_t code[C];
#pragma unroll
                    for (int r = 0; r < C; ++r)
                    {
                        const uint32_t e = E[r];
                        const uint32_t h = pk_max3h(pk_addh(diag, sc[r]), e, f);
                        diag = Hin[r];
                        Hout[r] = h;
                        const uint32_t tt = pk_addh_s(h, NEG6);                      // h - gapoe
                        const uint32_t en = pk_maxh(pk_dech(e), tt);                 // E of the next column
                        const uint32_t fn = pk_maxh(pk_dech(f), tt);                 // F of the next row
                        // 0 / 1 flags as f16 numbers: clamp(x - y) with x >= y integers
                        uint32_t ne, nf, xe, xf, c;
                        asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(ne) : "v"(h), "v"(e));
                        asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(nf) : "v"(h), "v"(f));
                        asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(xe) : "v"(en), "v"(tt));
                        asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(xf) : "v"(fn), "v"(tt));
                        asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(c) : "v"(xf), "v"(EIGHT2), "v"(B1024));  // 1024 + 8 xf
                        asm("v_pk_fma_f16 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]" : "=v"(c) : "v"(xe), "v"(c));
                        asm("v_pk_fma_f16 %0, %1, 2.0, %2 op_sel_hi:[1,0,1]" : "=v"(c) : "v"(nf), "v"(c));
                        code[r] = pk_addh(c, ne);
                        E[r] = en;
                        f = fn;
                    }
                    Fsend = f;
#pragma unroll
                    for (int r = 0; r < C; r += 2)
                        zw[((size_t)t * ZDW + (size_t)(r / 2)) * 64 + (size_t)lane] = __builtin_amdgcn_perm(code[r + 1], code[r], 0x06020400u);
                }
            };
            for (uint32_t t = 0; t < nsteps; t += 2)
            {
                step(HA, HB, sA, sB, t);
                step(HB, HA, sB, sA, t + 1);
            }
        }
        __threadfence_block();
        __syncthreads();

        // ---- traceback (ksw.c:513-528).  Eight lanes per candidate (lanes 0-7 of a read: A, 8-15: B) walk it together: in the
        // H state they look at the next eight cells of the diagonal at once and take the whole run of matches (one round trip
        // to memory instead of eight); gap states step one cell at a time.  Every lane of the eight follows the same state;
        // the first one writes the CIGAR.
        {
            const int h = k >> 3, d = k & 7;
            const FinishInfo& f = h == 0 ? gA : gB;
            const bool live = f.item != PG_NONE;
            const uint32_t w = base + (uint32_t)(2 * grp + h);
            // the candidate's short slot; `cap` entries from slot_base on, filled from the end (the walk runs backwards)
            uint32_t slot_base = w * a.cig_small, cap = a.cig_small;
            uint32_t* slot = a.cigars + slot_base;
            // the first lane of the eight writes: it moves the CIGAR into a full-size slot of the pool when the short one is full
            auto room = [&](uint32_t n_have) -> bool {
                if (n_have < cap)
                    return true;
                if (cap == a.cig_cap)
                    return false;
                if (d == 0)
                {
                    const uint32_t o = atomicAdd(a.ovf_count, 1u);
                    if (o >= a.ovf_cap)
                        return false;
                    const uint32_t nb = a.ovf_base + o * a.cig_cap;
                    uint32_t* ns = a.cigars + nb;
                    for (uint32_t e = 0; e < n_have; ++e)
                        ns[a.cig_cap - 1 - e] = slot[cap - 1 - e];
                    slot = ns;
                    slot_base = nb;
                }
                cap = a.cig_cap;
                return true;
            };
            const uint8_t* zb = (const uint8_t*)zw;
            const int ql = f.qe - f.qb + 1, tl = f.te - f.tb + 1;
            uint32_t n = 0, cur = 0;
            bool have = false, overflow = false;
            auto push = [&](uint32_t op, uint32_t len) {
                if (have && (cur & 0xfu) == op)
                    cur += len << 4;
                else
                {
                    if (have)
                    {
                        if (!overflow && room(n))
                        {
                            if (d == 0)
                                slot[cap - 1 - n] = cur;
                        }
                        else
                            overflow = true;
                        ++n;
                    }
                    cur = (len << 4) | op;
                    have = true;
                }
            };
            auto zbyte = [&](int ci, int cj) -> uint32_t {
                const int kk = cj / C, rr = cj % C;
                return zb[(((size_t)(ci + kk) * ZDW + (size_t)(rr / 2)) * 64 + (size_t)(grp * 16 + kk)) * 4 + (size_t)(h * 2 + (rr & 1))];
            };
            int i = live ? tl - 1 : -1, j = live ? ql - 1 : -1;
            uint32_t which = 0;
            const int sub_shift = grp * 16 + h * 8;
            while (__any(i >= 0 && j >= 0))
            {
                const bool on = i >= 0 && j >= 0;
                if (which == 0)
                {
                    const int ci = i - d, cj = j - d;
                    const bool inb = on && ci >= 0 && cj >= 0;
                    const uint32_t b = inb ? zbyte(ci, cj) : 0u;
                    const unsigned long long bal = __ballot(inb && (b & 3u) == 3u);
                    const uint32_t m8 = (uint32_t)(bal >> sub_shift) & 0xFFu;
                    const int m = __builtin_ctz(~m8);  // leading run of matches (0..8)
                    const uint32_t bm = (uint32_t)__shfl((int)b, (lane & ~7) | (m & 7));  // the cell that ends the run
                    if (on)
                    {
                        if (m > 0)
                        {
                            push(0, (uint32_t)m);
                            i -= m;
                            j -= m;
                        }
                        if (m < 8 && i >= 0 && j >= 0)
                        {
                            which = !(bm & 2u) ? 2u : 1u;  // not a match: F == H comes first, else E == H
                            if (which == 1u)
                            {
                                push(2, 1);
                                --i;
                            }
                            else
                            {
                                push(1, 1);
                                --j;
                            }
                        }
                    }
                }
                else if (on)
                {
                    const uint32_t b = zbyte(i, j);
                    which = which == 1u ? ((b & 4u) ? 1u : 0u) : ((b & 8u) ? 2u : 0u);
                    if (which == 0u)
                    {
                        push(0, 1);
                        --i;
                        --j;
                    }
                    else if (which == 1u)
                    {
                        push(2, 1);
                        --i;
                    }
                    else
                    {
                        push(1, 1);
                        --j;
                    }
                }
            }
            if (live && d == 0)
            {
                if (i >= 0)
                    push(2, (uint32_t)(i + 1));
                if (j >= 0)
                    push(1, (uint32_t)(j + 1));
                if (have)
                {
                    if (!overflow && room(n))
                        slot[cap - 1 - n] = cur;
                    else
                        overflow = true;
                    ++n;
                }
                KlibItem out = a.items[f.item];
                if (overflow)
                {
                    atomicOr(a.error, 2u);
                    out.valid = 0u;
                }
                else
                {
                    out.tb = f.tb;
                    out.qb = f.qb;
                    out.n_cigar = n;
                    out.cig_begin = slot_base + cap - n;
                    out.valid = f.te >= f.tb ? 1u : 0u;
                }
                a.items[f.item] = out;
            }
        }
    }
}

template <int C> hipError_t launch_local(const KlibArgs& a, uint32_t n_pairs, hipStream_t stream)
{
    const size_t lds = (size_t)PG_GROUPS * 4 * PG_GROUP_LANES * C * sizeof(uint32_t);
    if (lds > 48 * 1024)
    {
        const hipError_t e = hipFuncSetAttribute((const void*)pg_klib_local_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(pg_klib_local_kernel<C>, dim3(n_pairs), dim3(64), lds, stream, a);
    return hipGetLastError();
}
template <int C> size_t finish_lds()
{
    return (size_t)PG_GROUPS * 5 * PG_GROUP_LANES * C * sizeof(uint32_t) + (size_t)PG_GROUPS * finish_winp(C) * sizeof(uint16_t) + 8 * sizeof(FinishInfo);
}
template <int C> hipError_t launch_finish(const KlibArgs& a, uint32_t grid, hipStream_t stream)
{
    if (finish_lds<C>() > 48 * 1024)
    {
        const hipError_t e = hipFuncSetAttribute((const void*)pg_klib_finish_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)finish_lds<C>());
        if (e != hipSuccess)
            return e;
    }
    hipLaunchKernelGGL(pg_klib_finish_kernel<C>, dim3(grid), dim3(64), finish_lds<C>(), stream, a);
    return hipGetLastError();
}
}  // namespace

hipError_t pg_klib_launch_local(int C, const KlibArgs& a, uint32_t n_pairs, hipStream_t stream)
{
    if (n_pairs == 0)
        return hipSuccess;
    switch (C)
    {
    case 2: return launch_local<2>(a, n_pairs, stream);
    case 4: return launch_local<4>(a, n_pairs, stream);
    case 6: return launch_local<6>(a, n_pairs, stream);
    case 8: return launch_local<8>(a, n_pairs, stream);
    case 10: return launch_local<10>(a, n_pairs, stream);
    case 12: return launch_local<12>(a, n_pairs, stream);
    case 14: return launch_local<14>(a, n_pairs, stream);
    case 16: return launch_local<16>(a, n_pairs, stream);
    case 20: return launch_local<20>(a, n_pairs, stream);
    case 24: return launch_local<24>(a, n_pairs, stream);
    case 28: return launch_local<28>(a, n_pairs, stream);
    case 32: return launch_local<32>(a, n_pairs, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t pg_klib_launch_finish(int C, const KlibArgs& a, uint32_t grid, hipStream_t stream)
{
    if (grid == 0)
        return hipSuccess;
    switch (C)
    {
    case 2: return launch_finish<2>(a, grid, stream);
    case 4: return launch_finish<4>(a, grid, stream);
    case 6: return launch_finish<6>(a, grid, stream);
    case 8: return launch_finish<8>(a, grid, stream);
    case 10: return launch_finish<10>(a, grid, stream);
    case 12: return launch_finish<12>(a, grid, stream);
    case 14: return launch_finish<14>(a, grid, stream);
    case 16: return launch_finish<16>(a, grid, stream);
    case 20: return launch_finish<20>(a, grid, stream);
    case 24: return launch_finish<24>(a, grid, stream);
    case 28: return launch_finish<28>(a, grid, stream);
    case 32: return launch_finish<32>(a, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

// bytes of direction scratch one wavefront of the finish kernel needs
uint64_t pg_klib_finish_z_bytes(int C) { return (uint64_t)(finish_win(C) + 16 + 2) * (uint64_t)(C / 2) * 256u; }
