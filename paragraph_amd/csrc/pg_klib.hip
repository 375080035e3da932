// pg_klib.hip -- klib (ksw) stage of the aligner cascade on the device (--klib-sequence-matching).
//
// Replaces
//   grm::KlibAligner::{setGraph,alignRead}          src/c++/lib/grm/KlibAligner.cpp:186-205, 388-442
//   KlibAlignerImpl::{buildGraphCigar,pickBest}     src/c++/lib/grm/KlibAligner.cpp:207-308, 349-386
//   common::KlibAlignment::update / translate       src/c++/lib/common/Klib.cpp:144-164, KlibImpl.hh:77-102
//   ksw_i16 / ksw_align(KSW_XSTART) / ksw_global    external/klib/ksw.c:223-321, 330-355, 457-531
//   common::makeCigarBit                            src/c++/lib/common/Alignment.cpp:72-114
//
// Two kernels.
//  pg_klib_pair_kernel<R>: one WAVEFRONT per (read, path, strand) = one KlibAlignment::update().  Lane k owns R
//    consecutive query rows; the wave sweeps the target in skewed anti-diagonals (lane k works on column t-k at step
//    t), passing the last row's H and the running vertical gap F to lane k+1 with one cross-lane move per step, so the
//    whole DP state lives in VGPRs.  Three sweeps: (1) local affine SW over the whole path -> score and END cell
//    (first column holding the maximum; within it ksw's striped memory order: smallest row % slen, then row / slen);
//    (2) the same sweep over the reversed prefixes -> START cell (ksw_align's KSW_XSTOP pass); (3) ksw_global's
//    banded global DP over the local window in 32-bit, writing one direction byte per cell (coalesced 256 B per step)
//    that lane 0 then walks back into the run-length path CIGAR.  ksw_i16 is Farrar's striped kernel; cell by cell it
//    computes the textbook recurrence (its lazy-F shortcut only lowers stored E values that no H ever needs), which
//    is what sweep (1)/(2) evaluate; the claim is pinned against the reference's ksw.c by the CPU and GPU klib tests.
//  pg_klib_pick_kernel: one thread per read.  Replays the candidate heap (std::push_heap / pop_heap element order,
//    capacity paths + 2, worst score evicted), std::min_element pickBest, the equal-score "different (CIGAR, pos) ->
//    BAD_ALIGN" rule, and streams the graph CIGAR (path CIGAR split at node boundaries, M/X/N runs per piece).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"
#include "pg_internal.h"
#include "pg_klib.h"

namespace
{
constexpr int MAX_PATHS = PG_KLIB_MAX_PATHS_WIDE;  // the envelope; graph sets of up to PG_KLIB_MAX_PATHS paths per graph run the
                                                   // heap-replaying kernels with their heaps in registers
// KlibAlignerImpl's scoring (KlibAligner.cpp:134-142): ksw charges gapo + gape for the first gap base
constexpr int K_MATCH = 1, K_MISMATCH = -4, K_GAPO = 5, K_GAPE = 1, K_GAPOE = K_GAPO + K_GAPE;
constexpr int K_MINUS_INF = -0x40000000;
// direction bytes per lane per step: one per row, padded to a dword (R <= 4) or two (R <= 8)
constexpr int z_lane_bytes(int R) { return R <= 4 ? 4 : 8; }

__device__ __forceinline__ uint32_t comp_raw(uint32_t c) { return klib_comp_raw(c); }
__device__ __forceinline__ int ksw_code(uint32_t c) { return klib_code(c); }
__device__ __forceinline__ int ksw_score(int a, int b) { return ((a | b) & 4) ? 0 : (a == b ? K_MATCH : K_MISMATCH); }

struct ReadView
{
    const char* bases;
    int L;
    __device__ uint32_t at(int j, bool reverse) const { return reverse ? comp_raw((uint8_t)bases[L - 1 - j]) : (uint8_t)bases[j]; }
};

// value of lane - 1 (lane 0 gets `lane0`): one DPP move (wave_shr:1 works across the four 16-lane rows on gfx950;
// tools/ubench/dpp_wave_shr.hip), not a ds_bpermute round trip -- it sits on the step-to-step critical path
__device__ __forceinline__ int lane_up(int v, int lane0) { return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xf, 0xf, false); }

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        const unsigned long long o = __shfl_xor(v, m);
        v = o > v ? o : v;
    }
    return v;
}

// Local affine SW sweep (ksw_i16 semantics).  Rows = query rows qrow(j) for j in [0, nrows), columns = target codes
// tcode[t0 + tdir * i] for i in [0, ncols).  Returns score, the first column holding it and the row ksw picks there.
template <int R>
__device__ __forceinline__ void klib_local(
    const uint8_t* __restrict__ tcode, int t0, int tdir, int ncols, const int (&q)[R], int nrows, int lane, int& out_score, int& out_col,
    int& out_row)
{
    const int slen = (nrows + 7) / 8;
    const int nl = (nrows + R - 1) / R;
    int Hp[R], E[R], key[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
        Hp[r] = 0;
        E[r] = 0;
        const int row = lane * R + r;
        key[r] = (row % slen) * 8 + row / slen;
    }
    int my_hlast = 0, my_f = 0, diag0 = 0;
    // per row: its maximum and the first column reaching it (two selects per cell); the lane's (score, first column,
    // smallest key in that column) is put together from these after the sweep
    int row_h[R], row_col[R];
    bool row_ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
        row_h[r] = 0;
        row_col[r] = -1;
        row_ok[r] = lane * R + r < nrows;
    }
    const int steps = ncols + nl - 1;
    const bool lane_on = lane < nl;
    int tc_next = (lane_on && lane == 0 && ncols > 0) ? (int)tcode[t0] : 4;
    for (int t = 0; t < steps; ++t)
    {
        const int up_h = lane_up(my_hlast, 0);
        const int up_f = lane_up(my_f, 0);
        const int i = t - lane;
        const int tc = tc_next;
        {  // prefetch the next step's target code
            const int in = i + 1;
            tc_next = (lane_on && in >= 0 && in < ncols) ? (int)tcode[t0 + tdir * in] : 4;
        }
        if (lane_on && i >= 0 && i < ncols)
        {
            int d = diag0, f = up_f, h = 0;
#pragma unroll
            for (int r = 0; r < R; ++r)
            {
                const int s = ksw_score(tc, q[r]);
                int base = d + s;
                base = base > E[r] ? base : E[r];
                h = base > f ? base : f;
                d = Hp[r];
                Hp[r] = h;
                int o = h - K_GAPOE;
                o = o > 0 ? o : 0;
                int e = E[r] - K_GAPE;
                e = e > 0 ? e : 0;
                E[r] = e > o ? e : o;
                f -= K_GAPE;
                f = f > o ? f : o;
                const bool grew = row_ok[r] && h > row_h[r];
                row_h[r] = grew ? h : row_h[r];
                row_col[r] = grew ? i : row_col[r];
            }
            my_hlast = h;
            my_f = f;
            diag0 = up_h;
        }
    }
    int best_h = 0, best_col = -1, best_key = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
        best_h = row_h[r] > best_h ? row_h[r] : best_h;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (row_h[r] == best_h && best_h > 0 && (best_col < 0 || row_col[r] < best_col))
            best_col = row_col[r];
    best_key = 0x7FFFFFFF;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (row_h[r] == best_h && row_col[r] == best_col && key[r] < best_key)
            best_key = key[r];
    unsigned long long comp = 0;
    if (best_h > 0)
        comp = ((unsigned long long)best_h << 44) | ((unsigned long long)(0xFFFFF - best_col) << 24) | (unsigned long long)(0xFFFFFF - best_key);
    comp = wave_max_u64(comp);
    if (comp == 0)
    {
        out_score = 0;
        out_col = -1;
        out_row = 0;
        return;
    }
    out_score = (int)(comp >> 44);
    out_col = 0xFFFFF - (int)((comp >> 24) & 0xFFFFF);
    const int k = 0xFFFFFF - (int)(comp & 0xFFFFFF);
    out_row = k / 8 + (k % 8) * slen;
}

// ksw_global sweep over the window: rows q[0..nrows), columns tcode[t0 + i], band w.  Writes one direction byte per
// cell to z[((i + lane) * 64 + lane) * z_lane_bytes(R) + r].
template <int R>
__device__ __forceinline__ void klib_global(
    const uint8_t* __restrict__ tcode, int t0, int ncols, const int (&q)[R], int nrows, int w, int lane, uint8_t* __restrict__ z)
{
    const int nl = (nrows + R - 1) / R;
    int Hp[R], E[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
    {
        const int j = lane * R + r;
        Hp[r] = (j + 1 <= w) ? -(K_GAPO + K_GAPE * (j + 1)) : K_MINUS_INF;  // H[-1][j] (eh[j + 1].h, ksw.c:475-478)
        E[r] = K_MINUS_INF;
    }
    const int j0 = lane * R;
    int diag0 = lane == 0 ? 0 : ((j0 <= w) ? -(K_GAPO + K_GAPE * j0) : K_MINUS_INF);  // H[-1][j0 - 1]
    int my_hlast = K_MINUS_INF, my_f = K_MINUS_INF;
    const int steps = ncols + nl - 1;
    const bool lane_on = lane < nl;
    int tc_next = (lane_on && lane == 0 && ncols > 0) ? (int)tcode[t0] : 4;
    for (int t = 0; t < steps; ++t)
    {
        const int i = t - lane;
        // lane 0: H[i][-1] = -(gapo + gape * (i + 1)) (ksw.c:486, beg == 0; here i == t), F = -inf
        const int up_h = lane_up(my_hlast, -(K_GAPO + K_GAPE * (t + 1)));
        const int up_f = lane_up(my_f, K_MINUS_INF);
        const int tc = tc_next;
        {
            const int in = i + 1;
            tc_next = (lane_on && in >= 0 && in < ncols) ? (int)tcode[t0 + in] : 4;
        }
        if (lane_on && i >= 0 && i < ncols)
        {
            int d = diag0, f = up_f;
            uint64_t zw = 0;
#pragma unroll
            for (int r = 0; r < R; ++r)
            {
                const int j = j0 + r;
                if (j < nrows && j < i + w + 1)
                {
                    int h = d + ksw_score(tc, q[r]);
                    int e = E[r];
                    uint32_t dir = h > e ? 0u : 1u;
                    h = h > e ? h : e;
                    dir = h > f ? dir : 2u;
                    h = h > f ? h : f;
                    d = Hp[r];
                    Hp[r] = h;
                    h -= K_GAPOE;
                    e -= K_GAPE;
                    dir |= e > h ? 4u : 0u;
                    e = e > h ? e : h;
                    E[r] = e;
                    f -= K_GAPE;
                    dir |= f > h ? 32u : 0u;
                    f = f > h ? f : h;
                    zw |= (uint64_t)dir << (8 * r);
                }
                else
                {
                    d = Hp[r];
                    Hp[r] = K_MINUS_INF;
                    E[r] = K_MINUS_INF;
                }
            }
            if (R <= 4)
                *(uint32_t*)(z + ((size_t)t * 64 + (size_t)lane) * 4) = (uint32_t)zw;
            else
                *(uint64_t*)(z + ((size_t)t * 64 + (size_t)lane) * 8) = zw;
            my_hlast = Hp[R - 1];
            my_f = f;
            diag0 = up_h;
        }
    }
}

template <int R>
__global__ __launch_bounds__(64) void pg_klib_pair_kernel(KlibArgs a)
{
    const int lane = (int)threadIdx.x;
    const uint32_t per_read = 2u * a.max_paths;
    const uint64_t n_items = (uint64_t)a.n_reads * per_read;
    uint8_t* z = a.z + (size_t)blockIdx.x * a.z_bytes;
    for (uint64_t item = blockIdx.x; item < n_items; item += gridDim.x)
    {
        const uint32_t r = (uint32_t)(item / per_read), sub = (uint32_t)(item % per_read);
        const uint32_t pi = sub >> 1;
        const bool reverse = (sub & 1u) != 0;
        if (a.active && !a.active[r])
            continue;
        KlibItem out{};
        const uint32_t off = a.base_off[r];
        const int L = (int)(a.base_off[r + 1] - off);
        const LGraphDev g = a.graphs[a.graph_of_read[r]];
        if (pi >= g.n_paths || L == 0 || L > 64 * R)
        {
            if (lane == 0)
                a.items[item] = out;
            continue;
        }
        const LPathDev p = a.paths[g.path_off + pi];
        const uint8_t* tcode = a.pathcode + p.seq_off;
        ReadView rv{ a.bases + off, L };
        // ---- (1) forward local sweep
        int q[R];
#pragma unroll
        for (int k = 0; k < R; ++k)
        {
            const int j = lane * R + k;
            q[k] = j < L ? ksw_code(rv.at(j, reverse)) : 4;
        }
        int score, te, qe;
        klib_local<R>(tcode, 0, 1, (int)p.len, q, L, lane, score, te, qe);
        if (score <= 0)
        {  // ksw: te = -1, tb = 0 -> "fully soft clipped" (KlibAligner.cpp:404-408)
            if (lane == 0)
                a.items[item] = out;
            continue;
        }
        // ---- (2) reverse sweep over query[0..qe] x target[0..te], both reversed (ksw.c:347-352)
#pragma unroll
        for (int k = 0; k < R; ++k)
        {
            const int j = lane * R + k;
            q[k] = j <= qe ? ksw_code(rv.at(qe - j, reverse)) : 4;
        }
        int rscore, rte, rqe;
        klib_local<R>(tcode, te, -1, te + 1, q, qe + 1, lane, rscore, rte, rqe);
        const int tb = te - rte, qb = qe - rqe;
        if (rscore != score || tb < 0 || qb < 0)
        {  // cannot happen for the textbook recurrence; the reference's behaviour would be undefined (Klib.cpp:155-159)
            if (lane == 0)
                a.items[item] = out;
            continue;
        }
        // ---- (3) banded global alignment of the window (Klib.cpp:155-159: band = reflen)
        const int ql = qe - qb + 1, tl = te - tb + 1, w = (int)p.len;
#pragma unroll
        for (int k = 0; k < R; ++k)
        {
            const int j = lane * R + k;
            q[k] = j < ql ? ksw_code(rv.at(qb + j, reverse)) : 4;
        }
        klib_global<R>(tcode, tb, tl, q, ql, w, lane, z);
        __threadfence();
        __syncthreads();
        // ---- (4) backtrack (ksw.c:513-528) into the tail of this item's CIGAR slot
        if (lane == 0)
        {
            uint32_t* slot = a.cigars + item * a.cig_cap;
            uint32_t n = 0;       // entries written so far (tail-aligned)
            uint32_t cur = 0;     // entry being grown
            bool have = false, overflow = false;
            auto push = [&](uint32_t op, uint32_t len) {
                if (have && (cur & 0xfu) == op)
                    cur += len << 4;
                else
                {
                    if (have)
                    {
                        if (n < a.cig_cap)
                            slot[a.cig_cap - 1 - n] = cur;
                        else
                            overflow = true;
                        ++n;
                    }
                    cur = (len << 4) | op;
                    have = true;
                }
            };
            int i = tl - 1, k = (i + w + 1 < ql ? i + w + 1 : ql) - 1;
            uint32_t which = 0;
            while (i >= 0 && k >= 0)
            {
                const int ln = k / R, rr = k % R;
                const uint32_t zb = z[((size_t)(i + ln) * 64 + (size_t)ln) * z_lane_bytes(R) + (size_t)rr];
                which = (zb >> (which << 1)) & 3u;
                if (which == 0)
                {
                    push(0, 1);
                    --i;
                    --k;
                }
                else if (which == 1)
                {
                    push(2, 1);
                    --i;
                }
                else
                {
                    push(1, 1);
                    --k;
                }
            }
            if (i >= 0)
                push(2, (uint32_t)(i + 1));
            if (k >= 0)
                push(1, (uint32_t)(k + 1));
            if (have)
            {
                if (n < a.cig_cap)
                    slot[a.cig_cap - 1 - n] = cur;
                else
                    overflow = true;
                ++n;
            }
            if (overflow)
            {
                atomicOr(a.error, 2u);
                a.items[item] = out;
            }
            else
            {
                out.score = score;
                out.tb = tb;
                out.te = te;
                out.qb = qb;
                out.qe = qe;
                out.n_cigar = n;
                out.cig_begin = (uint32_t)(item * a.cig_cap) + a.cig_cap - n;
                out.valid = te >= tb ? 1u : 0u;
                a.items[item] = out;
            }
        }
        __syncthreads();  // the next item reuses z
    }
}

// Streams the tokens "(node, op, len)" of one candidate's graph CIGAR (buildGraphCigar, KlibAligner.cpp:207-308).
struct KGen
{
    const KlibArgs& a;
    LPathDev p;
    ReadView rv;
    bool reverse;
    const uint32_t* cig;  // ksw_global entries
    uint32_t n_cig;
    uint32_t left, right;  // soft clips
    // state
    int elem;  // -1 = left clip pending, 0..n_cig-1, n_cig = right clip, n_cig+1 = done
    uint32_t remaining;  // of the current ALIGN / DELETE element
    uint32_t cur_op;     // ksw op of the current element
    uint32_t piece;      // bases of the current ALIGN piece still to emit
    uint32_t node_idx, node_pos, node_first;
    int it;
    int32_t graph_pos;

    __device__ uint32_t ref(uint32_t i) const { return (uint8_t)a.pathseq[p.seq_off + i]; }
    __device__ uint32_t start_of(uint32_t i) const { return a.starts[p.start_off + 2 * i]; }
    __device__ uint32_t node_of(uint32_t i) const { return a.starts[p.start_off + 2 * i + 1]; }
    __device__ uint32_t node_len(uint32_t i) const { return (i + 1 < p.n_nodes ? start_of(i + 1) : p.len) - start_of(i); }

    __device__ void init(const KlibItem& ki, const uint32_t* cigars, int L)
    {
        cig = cigars + ki.cig_begin;
        n_cig = ki.n_cigar;
        left = (uint32_t)ki.qb;
        right = (uint32_t)(L - ki.qe - 1);
        node_idx = 0;
        for (uint32_t i = 0; i < p.n_nodes; ++i)
            if (start_of(i) <= (uint32_t)ki.tb)
                node_idx = i;
        node_first = start_of(node_idx);
        node_pos = (uint32_t)ki.tb - node_first;
        graph_pos = (int32_t)node_pos;
        elem = left ? -1 : 0;
        remaining = 0;
        piece = 0;
        cur_op = 0;
        it = 0;
    }

    __device__ bool next(uint32_t& node, uint32_t& op, uint32_t& len)
    {
        for (;;)
        {
            if (piece)
            {  // one run of equal ops inside the current ALIGN piece (makeCigarBit)
                const uint32_t r0 = node_first + node_pos;
                auto opat = [&](uint32_t j) -> uint32_t {
                    const uint32_t rc = ref(r0 + j), qc = rv.at(it + (int)j, reverse);
                    return rc == qc ? PG_OPC_M : ((rc == 'N' || qc == 'N') ? PG_OPC_N : PG_OPC_X);
                };
                const uint32_t o = opat(0);
                uint32_t run = 1;
                while (run < piece && opat(run) == o)
                    ++run;
                node = node_of(node_idx);
                op = o;
                len = run;
                it += (int)run;
                node_pos += run;
                piece -= run;
                return true;
            }
            if (remaining)
            {  // next node piece of an ALIGN / DELETE element
                uint32_t room = node_len(node_idx) - node_pos;
                if (room == 0)
                {
                    node_first += node_len(node_idx);
                    ++node_idx;
                    node_pos = 0;
                    if (node_idx >= p.n_nodes)
                        return false;  // cannot happen: the alignment lies inside the path
                    room = node_len(node_idx);
                }
                const uint32_t take = remaining < room ? remaining : room;
                remaining -= take;
                if (cur_op == 0)
                {
                    piece = take;
                    continue;
                }
                node = node_of(node_idx);
                op = PG_OPC_D;
                len = take;
                node_pos += take;
                return true;
            }
            // next element
            if (elem == -1)
            {
                elem = 0;
                node = node_of(node_idx);
                op = PG_OPC_S;
                len = left;
                it += (int)left;
                return true;
            }
            if ((uint32_t)elem < n_cig)
            {
                const uint32_t v = cig[elem++];
                cur_op = v & 0xfu;
                const uint32_t l = v >> 4;
                if (cur_op == 1)
                {
                    node = node_of(node_idx);
                    op = PG_OPC_I;
                    len = l;
                    it += (int)l;
                    return true;
                }
                remaining = l;
                continue;
            }
            if ((uint32_t)elem == n_cig)
            {
                ++elem;
                if (right)
                {
                    node = node_of(node_idx);
                    op = PG_OPC_S;
                    len = right;
                    it += (int)right;
                    return true;
                }
            }
            return false;
        }
    }
};

struct HeapEnt
{
    int32_t score;
    uint32_t item;  // index inside the read's item block
};

__device__ void kheap_push(HeapEnt* h, int& n, const HeapEnt& v)
{  // std::push_heap with Candidate::betterScore as "less" (left.score > right.score)
    int hole = n;
    ++n;
    int parent = (hole - 1) / 2;
    while (hole > 0 && h[parent].score > v.score)
    {
        h[hole] = h[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h[hole] = v;
}

__device__ void kheap_pop(HeapEnt* h, int& n)
{  // std::pop_heap + pop_back (libstdc++ __adjust_heap)
    const HeapEnt value = h[n - 1];
    const int len = n - 1;
    int hole = 0, second = 0;
    while (second < (len - 1) / 2)
    {
        second = 2 * (second + 1);
        if (h[second].score > h[second - 1].score)
            --second;
        h[hole] = h[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2)
    {
        second = 2 * (second + 1);
        h[hole] = h[second - 1];
        hole = second - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > 0 && h[parent].score > value.score)
    {
        h[hole] = h[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    if (len > 0)
        h[hole] = value;
    n = len;
}

// Replays a read's candidate heap on the scores of the first pass (KlibAligner.cpp:388-442: push, evict the worst beyond
// paths + 2) and puts the candidates the pick kernel can look at -- those holding the best score -- on the work list of the
// finish kernel.
template <int HEAP_CAP> __global__ __launch_bounds__(64) void pg_klib_select_kernel(KlibArgs a)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= a.n_reads || (a.active && !a.active[r]))
        return;
    if (a.base_off[r + 1] == a.base_off[r] || a.base_off[r + 1] - a.base_off[r] > a.len_limit)
        return;
    const LGraphDev g = a.graphs[a.graph_of_read[r]];
    if (g.n_paths == 0 || (int)g.n_paths + 2 > HEAP_CAP)
        return;
    const uint32_t per_read = 2u * a.max_paths;
    const uint64_t item0 = (uint64_t)r * per_read;
    HeapEnt heap[HEAP_CAP];
    int hn = 0;
    const int cap = (int)g.n_paths + 2;
    for (uint32_t s = 0; s < 2u * g.n_paths; ++s)
    {
        const KlibItem ki = a.items[item0 + s];
        if (!ki.valid)
            continue;
        kheap_push(heap, hn, HeapEnt{ ki.score, s });
        if (hn == cap)
            kheap_pop(heap, hn);
    }
    int best = 0;
    for (int i = 0; i < hn; ++i)
        best = heap[i].score > best ? heap[i].score : best;
    for (int i = 0; i < hn; ++i)
        if (heap[i].score == best)
            a.worklist[atomicAdd(a.work_count, 1u)] = (uint32_t)(item0 + heap[i].item);
}

template <int HEAP_CAP> __global__ __launch_bounds__(64) void pg_klib_pick_kernel(KlibArgs a)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= a.n_reads || (a.active && !a.active[r]))
        return;
    a.flags[r] = 0;
    const uint32_t off = a.base_off[r];
    const int L = (int)(a.base_off[r + 1] - off);
    if (L == 0 || (uint32_t)L > a.len_limit)
        return;
    const LGraphDev g = a.graphs[a.graph_of_read[r]];
    if (g.n_paths == 0 || (int)g.n_paths + 2 > HEAP_CAP)
        return;
    ReadView rv{ a.bases + off, L };
    const uint32_t per_read = 2u * a.max_paths;
    const uint64_t item0 = (uint64_t)r * per_read;
    HeapEnt heap[HEAP_CAP];
    int hn = 0;
    const int cap = (int)g.n_paths + 2;
    for (uint32_t s = 0; s < 2u * g.n_paths; ++s)
    {
        const KlibItem ki = a.items[item0 + s];
        if (!ki.valid)
            continue;
        kheap_push(heap, hn, HeapEnt{ ki.score, s });
        if (hn == cap)
            kheap_pop(heap, hn);
    }
    if (hn == 0)
        return;
    // ---- pickBest (KlibAligner.cpp:349-386): std::min_element with betterScore = first best score
    int bi = 0;
    for (int i = 1; i < hn; ++i)
        if (heap[i].score > heap[bi].score)
            bi = i;
    const HeapEnt best = heap[bi];
    const KlibItem kb = a.items[item0 + best.item];
    const LPathDev pb = a.paths[g.path_off + (best.item >> 1)];
    const bool rev_b = (best.item & 1u) != 0;
    bool bad = false;
    {
        int i = bi + 1;
        while (i < hn)
        {
            int si = i;
            for (int j = i + 1; j < hn; ++j)
                if (heap[j].score > heap[si].score)
                    si = j;
            const HeapEnt sb = heap[si];
            if (sb.score != best.score)
                break;
            const KlibItem ks = a.items[item0 + sb.item];
            KGen g1{ a, pb, rv, rev_b };
            KGen g2{ a, a.paths[g.path_off + (sb.item >> 1)], rv, (sb.item & 1u) != 0 };
            g1.init(kb, a.cigars, L);
            g2.init(ks, a.cigars, L);
            bool differ = g1.graph_pos != g2.graph_pos;
            while (!differ)
            {
                uint32_t n1, o1, l1, n2, o2, l2;
                const bool h1 = g1.next(n1, o1, l1), h2 = g2.next(n2, o2, l2);
                if (h1 != h2)
                    differ = true;
                else if (!h1)
                    break;
                else if (n1 != n2 || o1 != o2 || l1 != l2)
                    differ = true;
            }
            if (differ)
            {
                bad = true;
                break;
            }
            i = si + 1;
        }
    }
    // ---- emit the best alignment
    KGen gen{ a, pb, rv, rev_b };
    gen.init(kb, a.cigars, L);
    uint32_t n_ops = 0, score = 0, clipped = 0;
    {
        uint32_t nd, op, len;
        while (gen.next(nd, op, len))
            ++n_ops;
    }
    const unsigned long long base = atomicAdd(a.ops_counter, (unsigned long long)n_ops);
    if (base + n_ops > a.ops_cap)
    {
        atomicOr(a.error, 1u);
        return;
    }
    gen.init(kb, a.cigars, L);
    {
        uint32_t nd, op, len, e = 0;
        while (gen.next(nd, op, len))
        {
            a.ops[base + e++] = PG_OP_MAKE(nd, op, len);
            if (op == PG_OPC_M)
                score += len;
            if (op == PG_OPC_S)
                clipped += len;
        }
    }
    pg_result res;
    res.graph_pos = gen.graph_pos;
    res.score = (int16_t)score;
    res.mapq = bad ? 0 : 60;
    res.is_unique = bad ? 0 : 1;
    res.returned_reverse = (uint8_t)rev_b;
    res.multi_mask = 0;
    res.n_ops = (uint16_t)n_ops;
    res.ops_off = (uint32_t)base;
    res.strand_score[0] = rev_b ? -1 : (int16_t)kb.score;
    res.strand_score[1] = rev_b ? (int16_t)kb.score : -1;
    res.clipped = (uint16_t)clipped;
    res.status = PG_STATUS_KLIB_ALIGNER;
    a.results[r] = res;
    a.flags[r] = bad ? 4 : 1;  // bit0 MAPPED, bit2 BAD_ALIGN (ambiguous best)
}
}  // namespace

struct pg_klib_index
{
    uint32_t max_paths = 0;     // per graph
    uint32_t max_path_len = 0, min_path_len = 0;
    LGraphDev* d_graphs = nullptr;
    uint32_t* d_pathmeta = nullptr;  // packed kernels: one word per path column (its code) + PG_META_PAD idle words per path
    uint32_t* d_worklist = nullptr;
    size_t worklist_cap = 0;
    uint32_t* d_work_count = nullptr;
    uint32_t last_packed = 0;
    LPathDev* d_paths = nullptr;
    char* d_pathseq = nullptr;
    uint8_t* d_pathcode = nullptr;
    uint32_t* d_starts = nullptr;
    // stage scratch (grown on demand)
    KlibItem* d_items = nullptr;
    size_t items_cap = 0;
    uint32_t* d_cigars = nullptr;
    size_t cigars_cap = 0;
    uint8_t* d_z = nullptr;
    size_t z_cap = 0;
    uint32_t* d_error = nullptr;
};

void pg_klib_index_free(pg_klib_index* ix)
{
    if (!ix)
        return;
    (void)pg_dev_free(ix->d_graphs);
    (void)pg_dev_free(ix->d_paths);
    (void)pg_dev_free(ix->d_pathseq);
    (void)pg_dev_free(ix->d_pathcode);
    (void)pg_dev_free(ix->d_starts);
    (void)pg_dev_free(ix->d_pathmeta);
    (void)pg_dev_free(ix->d_worklist);
    (void)pg_dev_free(ix->d_work_count);
    (void)pg_dev_free(ix->d_items);
    (void)pg_dev_free(ix->d_cigars);
    (void)pg_dev_free(ix->d_z);
    (void)pg_dev_free(ix->d_error);
    delete ix;
}

template <typename T> static hipError_t upl(const std::vector<T>& v, T** d, hipStream_t s)
{
    hipError_t e = pg_dev_alloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(T));
    if (e != hipSuccess || v.empty())
        return e;
    return hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
}

extern "C" pg_status pg_graphs_build_klib_index(
    pg_ctx* ctx, pg_graphs* G, const uint32_t* path_off, const uint32_t* path_node_off, const uint32_t* path_nodes)
{
    if (!ctx || !G || !path_off || !path_node_off)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_graphs_build_klib_index: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<LGraphDev> gd(G->n_graphs);
    std::vector<LPathDev> pd;
    std::vector<char> pathseq;
    std::vector<uint8_t> pathcode;
    std::vector<uint32_t> starts;
    uint32_t max_len = 0, max_paths = 0, min_len = 0xFFFFFFFFu;
    size_t meta_words = 0;
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nb = G->h_node_off[g], n_nodes = G->h_node_off[g + 1] - nb;
        gd[g].path_off = path_off[g];
        gd[g].n_paths = path_off[g + 1] - path_off[g];
        if (gd[g].n_paths > (uint32_t)MAX_PATHS)
            return pg_fail(ctx, PG_ERR_UNSUPPORTED, "more than 126 paths on one graph");
        max_paths = std::max(max_paths, gd[g].n_paths);
        for (uint32_t p = path_off[g]; p < path_off[g + 1]; ++p)
        {
            LPathDev lp{};
            lp.seq_off = (uint32_t)pathseq.size();
            lp.start_off = (uint32_t)starts.size();
            lp.n_nodes = path_node_off[p + 1] - path_node_off[p];
            uint32_t pos = 0;
            for (uint32_t q = path_node_off[p]; q < path_node_off[p + 1]; ++q)
            {
                const uint32_t node = path_nodes[q];
                if (node >= n_nodes)
                    return pg_fail(ctx, PG_ERR_INVALID, "path node id out of range");
                const uint32_t so = G->h_nodeseq_off[nb + node], len = G->h_node_len[nb + node];
                if (len == 0)
                    return pg_fail(ctx, PG_ERR_UNSUPPORTED, "empty node on a path");
                starts.push_back(pos);
                starts.push_back(node);
                pathseq.insert(pathseq.end(), G->h_seq_raw.begin() + so, G->h_seq_raw.begin() + so + len);
                pos += len;
            }
            lp.len = pos;
            if (pos >= (1u << 20))
                return pg_fail(ctx, PG_ERR_UNSUPPORTED, "path longer than 2^20 bases");
            max_len = std::max(max_len, pos);
            min_len = std::min(min_len, pos);
            lp.meta_off = (uint32_t)meta_words;
            meta_words += (size_t)pos + PG_META_PAD;
            if (meta_words >= (1ull << 32))
                return pg_fail(ctx, PG_ERR_UNSUPPORTED, "more than 2^32 path bases in one graph set");
            pd.push_back(lp);
        }
    }
    pathcode.resize(pathseq.size());
    for (size_t i = 0; i < pathseq.size(); ++i)
    {
        switch (((int)pathseq[i]) & 0x7f)
        {
        case 'A': case 'a': case 'U': case 'u': pathcode[i] = 0; break;
        case 'C': case 'c': pathcode[i] = 1; break;
        case 'G': case 'g': pathcode[i] = 2; break;
        case 'T': case 't': pathcode[i] = 3; break;
        default: pathcode[i] = 4;
        }
    }
    std::vector<uint32_t> pathmeta(meta_words, PG_META_IDLE);
    for (const LPathDev& lp : pd)
        for (uint32_t i = 0; i < lp.len; ++i)
            pathmeta[lp.meta_off + i] = pathcode[lp.seq_off + i];
    pg_klib_index* ix = new pg_klib_index();
    ix->max_paths = max_paths;
    ix->max_path_len = max_len;
    ix->min_path_len = pd.empty() ? 0u : min_len;
    hipError_t e = upl(gd, &ix->d_graphs, ctx->stream_copy);
    if (e == hipSuccess) e = upl(pd, &ix->d_paths, ctx->stream_copy);
    if (e == hipSuccess) e = upl(pathseq, &ix->d_pathseq, ctx->stream_copy);
    if (e == hipSuccess) e = upl(pathcode, &ix->d_pathcode, ctx->stream_copy);
    if (e == hipSuccess) e = upl(starts, &ix->d_starts, ctx->stream_copy);
    if (e == hipSuccess) e = upl(pathmeta, &ix->d_pathmeta, ctx->stream_copy);
    if (e == hipSuccess) e = pg_dev_alloc((void**)&ix->d_work_count, 2 * sizeof(uint32_t));  // [0] work list, [1] CIGAR pool
    if (e == hipSuccess) e = pg_dev_alloc((void**)&ix->d_error, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(ix->d_error, 0, sizeof(uint32_t), ctx->stream_copy);
    if (e == hipSuccess) e = pg_stream_wait(ctx->device, ctx->stream_copy);
    if (e != hipSuccess)
    {
        pg_klib_index_free(ix);
        return pg_fail(ctx, PG_ERR_HIP, std::string("klib index upload: ") + hipGetErrorString(e));
    }
    pg_klib_index_free(G->klib_index);
    G->klib_index = ix;
    return PG_OK;
}

// slot of pg_ctx::klib_scratch, grown (never shrunk) to `need` elements; a block in use by an earlier stage call is waited for
template <typename T> static pg_status scratch(pg_ctx* ctx, int slot, T** d, size_t need)
{
    const size_t bytes = std::max<size_t>(need, 1) * sizeof(T);
    if (bytes > ctx->klib_scratch_bytes[slot])
    {
        if (ctx->klib_scratch[slot])
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)pg_dev_free(ctx->klib_scratch[slot]);
        ctx->klib_scratch[slot] = nullptr;
        ctx->klib_scratch_bytes[slot] = 0;
        const size_t cap = bytes + bytes / 4;
        HIP_TRY(ctx, pg_dev_alloc(&ctx->klib_scratch[slot], cap));
        ctx->klib_scratch_bytes[slot] = cap;
    }
    *d = (T*)ctx->klib_scratch[slot];
    return PG_OK;
}

template <typename T> static pg_status grow(pg_ctx* ctx, T** d, size_t* cap, size_t need)
{
    if (need <= *cap)
        return PG_OK;
    if (*d)  // (a block of an earlier call may still be read; a fresh index -- every batch of a workflow -- has none and waits for nothing)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    (void)pg_dev_free(*d);
    *d = nullptr;
    *cap = 0;
    HIP_TRY(ctx, pg_dev_alloc((void**)d, need * sizeof(T)));
    *cap = need;
    return PG_OK;
}

extern "C" pg_status pg_batch_klib_align(pg_ctx* ctx, pg_batch* b, uint32_t flags)
{
    if (!ctx || !b || !b->graphs)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_klib_align: batch not uploaded");
    const pg_graphs* G = b->graphs;
    if (!G->klib_index)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_klib_align: call pg_graphs_build_klib_index first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pg_klib_index* ix = G->klib_index;
    // reads beyond the stage's 512 bases are left alone (no result, no flag): in the cascade they fall through to the graph
    // aligner, whose general path takes them -- one long read must not cost the stage its batch
    constexpr uint32_t KLIB_LEN_LIMIT = 512;
    uint32_t max_len = 0;
    for (uint32_t r = 0; r < b->n_reads; ++r)
    {
        const uint32_t L = b->h_base_off[r + 1] - b->h_base_off[r];
        if (L <= KLIB_LEN_LIMIT)
            max_len = std::max(max_len, L);
    }
    const int R = std::max(1, (int)((max_len + 63) / 64));
    const uint64_t n_items = (uint64_t)b->n_reads * 2u * ix->max_paths;
    b->h_counters_valid = false;
    b->seed_chain = false;
    {
        const pg_status cp = pg_cascade_prepare_early(ctx, b);  // (a hand-over behind this stage finds its tables on the device)
        if (cp != PG_OK)
            return cp;
    }
    HIP_TRY(ctx, pg_stage_begin(ctx, b));
    // every way out of this call from here on records the end of the stage: pg_graphs_destroy / pg_dev_free trust those events
    // (an early return that skipped it could hand buffers the queued kernels still read to another lane)
    struct StageEnd
    {
        pg_ctx* c;
        pg_batch* b;
        bool done;
        ~StageEnd()
        {
            if (!done)
                (void)pg_stage_end(c, b);
        }
    } stage_end{ ctx, b, false };
    {
        const pg_status ps = pg_batch_ensure_plan(ctx, b, ctx->stream);  // the packed kernels run the batch's work items
        if (ps != PG_OK)
            return ps;
    }
    if (!(flags & PG_AF_KEEP_RESULTS) || flags == PG_AF_ALL)
        if (!b->ops_counter_fresh)
            HIP_TRY(ctx, hipMemsetAsync(b->d_ops_counter, 0, sizeof(unsigned long long), ctx->stream));
    b->ops_counter_fresh = false;
    if (b->n_reads == 0 || n_items == 0)
        return PG_OK;
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const uint32_t cig_cap = 2 * max_len + 4;
    // The packed kernels take reads on paths
    // no shorter than a read (ksw_global's band = path length then covers the whole window) and short enough for a 16-bit
    // step counter; anything else runs on the general kernels.
    const bool packed = ix->min_path_len >= max_len && ix->max_path_len <= 65000u && !std::getenv("PG_KLIB_GENERAL");
    ix->last_packed = packed ? 1u : 0u;
    KlibArgs a{};
    a.n_reads = b->n_reads;
    a.max_paths = ix->max_paths;
    a.base_off = b->d_base_off;
    a.bases = b->d_bases;
    a.graph_of_read = b->d_graph_of_read;
    a.graphs = ix->d_graphs;
    a.paths = ix->d_paths;
    a.pathseq = ix->d_pathseq;
    a.pathcode = ix->d_pathcode;
    a.pathmeta = ix->d_pathmeta;
    a.starts = ix->d_starts;
    a.active = b->has_active ? b->d_active : nullptr;
    a.len_limit = KLIB_LEN_LIMIT;
    a.cig_cap = cig_cap;
    a.results = b->d_results;
    a.ops = b->d_ops;
    a.ops_counter = b->d_ops_counter;
    a.ops_cap = b->ops_cap;
    a.flags = b->d_path_flags;
    a.error = ix->d_error;
    KlibItem* d_items = nullptr;
    uint32_t *d_worklist = nullptr, *d_cigars = nullptr;
    uint8_t* d_z = nullptr;
    pg_status st = scratch(ctx, 0, &d_items, (size_t)n_items);
    if (st != PG_OK)
        return st;
    a.items = d_items;
    if (packed)
    {
        st = scratch(ctx, 1, &d_worklist, (size_t)n_items);
        if (st != PG_OK)
            return st;
        a.worklist = d_worklist;
        a.work_count = ix->d_work_count;
        a.work = b->d_items;
        HIP_TRY(ctx, hipMemsetAsync(ix->d_work_count, 0, 2 * sizeof(uint32_t), ctx->stream));
        // first pass: the batch's wavefront work items (4 reads of one graph and one length class each), every path of the graph
        for (const Chunk& c : b->chunks)
        {
            a.pair_begin = c.pair_begin;
            HIP_TRY(ctx, pg_klib_launch_local(pg_var_c(c.C), a, c.pair_end - c.pair_begin, ctx->stream));
        }
        if (ix->max_paths <= (uint32_t)PG_KLIB_MAX_PATHS)
            hipLaunchKernelGGL(pg_klib_select_kernel<PG_KLIB_MAX_PATHS + 2>, dim3((b->n_reads + 63) / 64), dim3(64), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(pg_klib_select_kernel<PG_KLIB_MAX_PATHS_WIDE + 2>, dim3((b->n_reads + 63) / 64), dim3(64), 0, ctx->stream, a);
        HIP_TRY(ctx, hipGetLastError());
        // The size of the second pass is known on the DEVICE only: the finish kernel reads it there, the launch is sized for the
        // resident wavefronts and the scratch for every candidate.  (A read-back here was a wait for the whole main stream -- the
        // fills of every batch queued before this one -- under the caller's device lock: the workflow with the klib stage ran at
        // 5 - 27 k sites/s.)
        // CIGAR scratch: a short slot per candidate (PG_KLIB_CIG_SMALL entries; a CIGAR of the best score is a handful of runs)
        // + a pool of full-size slots for the ones that outgrow theirs, one per 32 candidates and at least 4 096: 150 bp reads on
        // ten paths cost 0.13 KB of scratch per candidate, not 1.2 KB, and the 32-bit entry index holds 100 M candidates.
        // PG_KLIB_CIG_POOL=<slots> sizes the pool for batches of reads with many indels (an exhausted pool is error bit 1 of
        // pg_graphs_klib_error, never a wrong CIGAR).
        if (n_items >= (1ull << 32))
            return pg_fail(ctx, PG_ERR_UNSUPPORTED, "pg_batch_klib_align: batch too large (candidate index)");
        const uint32_t n_work = (uint32_t)n_items;
        const uint32_t cig_small = std::min<uint32_t>(cig_cap, PG_KLIB_CIG_SMALL);
        uint64_t ovf_cap = cig_small == cig_cap ? 0u : std::min<uint64_t>(n_work, std::max<uint64_t>(4096u, n_work / 32u));
        if (const char* e = std::getenv("PG_KLIB_CIG_POOL"))
            ovf_cap = cig_small == cig_cap ? 0u : std::min<uint64_t>(n_work, std::strtoull(e, nullptr, 10));
        const uint64_t cig_entries = (uint64_t)n_work * cig_small + ovf_cap * cig_cap;
        if (cig_entries >= (1ull << 32))
            return pg_fail(ctx, PG_ERR_UNSUPPORTED, "pg_batch_klib_align: batch too large (CIGAR scratch index)");
        const int C = pg_var_c(pg_variant_of(max_len));
        const uint32_t waves = (n_work + 7u) / 8u;  // (an upper bound: every candidate)
        // resident wavefronts per CU: 10 by LDS for reads up to 250 bases, 4 (one per SIMD, by registers) beyond; each one owns
        // z_bytes of direction scratch
        const uint32_t grid = std::min<uint32_t>(waves, (uint32_t)n_cu * (C > 16 ? 4u : 10u));
        const uint64_t z_bytes = pg_klib_finish_z_bytes(C);
        st = scratch(ctx, 2, &d_cigars, std::max<size_t>((size_t)cig_entries, 1));
        if (st == PG_OK) st = scratch(ctx, 3, &d_z, std::max<size_t>((size_t)grid * z_bytes, 1));
        if (st != PG_OK)
            return st;
        a.cigars = d_cigars;
        a.cig_small = cig_small;
        a.ovf_cap = (uint32_t)ovf_cap;
        a.ovf_base = n_work * cig_small;
        a.ovf_count = ix->d_work_count + 1;
        a.z = d_z;
        a.z_bytes = z_bytes;
        a.n_work = 0;  // (unused: a.work_count is set)
        HIP_TRY(ctx, pg_klib_launch_finish(C, a, grid, ctx->stream));
    }
    else
    {
        if (n_items * cig_cap >= (1ull << 32))
            return pg_fail(ctx, PG_ERR_UNSUPPORTED, "pg_batch_klib_align: batch too large (CIGAR scratch index)");
        const uint32_t grid = (uint32_t)std::min<uint64_t>(n_items, (uint64_t)n_cu * 32u);
        // direction bytes: the window has at most 2 * L target columns (a local alignment with positive score cannot
        // delete more bases than it matches) and never more than the path; + 64 steps of skew
        const uint64_t z_steps = std::min<uint64_t>(2ull * max_len, ix->max_path_len) + 64 + 1;
        const uint64_t z_bytes = z_steps * 64 * z_lane_bytes(R);
        st = scratch(ctx, 2, &d_cigars, (size_t)n_items * cig_cap);
        if (st == PG_OK) st = scratch(ctx, 3, &d_z, (size_t)grid * z_bytes);
        if (st != PG_OK)
            return st;
        a.cigars = d_cigars;
        a.z = d_z;
        a.z_bytes = z_bytes;
        switch (R)
        {
        case 1: hipLaunchKernelGGL(pg_klib_pair_kernel<1>, dim3(grid), dim3(64), 0, ctx->stream, a); break;
        case 2: hipLaunchKernelGGL(pg_klib_pair_kernel<2>, dim3(grid), dim3(64), 0, ctx->stream, a); break;
        case 3: hipLaunchKernelGGL(pg_klib_pair_kernel<3>, dim3(grid), dim3(64), 0, ctx->stream, a); break;
        case 4: hipLaunchKernelGGL(pg_klib_pair_kernel<4>, dim3(grid), dim3(64), 0, ctx->stream, a); break;
        case 5: hipLaunchKernelGGL(pg_klib_pair_kernel<5>, dim3(grid), dim3(64), 0, ctx->stream, a); break;
        case 6: hipLaunchKernelGGL(pg_klib_pair_kernel<6>, dim3(grid), dim3(64), 0, ctx->stream, a); break;
        case 7: hipLaunchKernelGGL(pg_klib_pair_kernel<7>, dim3(grid), dim3(64), 0, ctx->stream, a); break;
        default: hipLaunchKernelGGL(pg_klib_pair_kernel<8>, dim3(grid), dim3(64), 0, ctx->stream, a); break;
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    if (ix->max_paths <= (uint32_t)PG_KLIB_MAX_PATHS)
        hipLaunchKernelGGL(pg_klib_pick_kernel<PG_KLIB_MAX_PATHS + 2>, dim3((b->n_reads + 63) / 64), dim3(64), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(pg_klib_pick_kernel<PG_KLIB_MAX_PATHS_WIDE + 2>, dim3((b->n_reads + 63) / 64), dim3(64), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    stage_end.done = true;
    HIP_TRY(ctx, pg_stage_end(ctx, b));
    return PG_OK;
}

/* bit0 = the batch's op buffer overflowed, bit1 = a path CIGAR exceeded its slot; clears the word. Synchronises. */
extern "C" pg_status pg_graphs_klib_error(pg_ctx* ctx, pg_graphs* G, uint32_t* error)
{
    if (!ctx || !G || !G->klib_index || !error)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_graphs_klib_error: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the last stage that used this graph set on the main stream (the klib stage), not everything every batch has queued there;
    // then the word comes down the copy stream
    if (G->use_recorded[0])
        HIP_TRY(ctx, hipEventSynchronize(G->ev_use[0]));
    HIP_TRY(ctx, hipMemcpyAsync(error, G->klib_index->d_error, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream_copy));
    HIP_TRY(ctx, hipMemsetAsync(G->klib_index->d_error, 0, sizeof(uint32_t), ctx->stream_copy));
    HIP_TRY(ctx, pg_stream_wait(ctx->device, ctx->stream_copy));
    return PG_OK;
}

extern "C" pg_status pg_graphs_klib_last_kernels(pg_ctx* ctx, pg_graphs* G, uint32_t* packed)
{
    if (!ctx || !G || !G->klib_index || !packed)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_graphs_klib_last_kernels: bad argument");
    *packed = G->klib_index->last_packed;
    return PG_OK;
}
