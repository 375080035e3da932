// pg_path.hip -- exact path matching stage of the aligner cascade on the device.
//
// Replaces
//   grm::PathAligner::{setGraph,alignRead}            src/c++/lib/grm/PathAligner.cpp:70-164
//   graphtools::KmerIndex (construction + numPaths)   GT!/src/graphalign/KmerIndex.cpp:76-116, 209-223
//   graphtools::extendPathMatching                    GT!/src/graphcore/PathOperations.cpp:117-271
//   projectAlignmentOntoGraph for an all-match alignment + GraphAlignment::generateCigar
//                                                     GT!/src/graphalign/GraphAlignmentOperations.cpp:130-164
//
// Host: enumerates every length-k path of every graph (depth-first over successors in ascending id, as
// extendPathEnd does), groups them by sequence and builds one open-addressing hash table per graph:
// key = 64-bit polynomial hash of the k raw characters, value = (number of paths with that sequence, first path).
// Device: sixteen lanes per read; both strands; all k-mer positions of a 128-position window hashed and probed in one round trip;
// for k-mers with exactly one path the reference's greedy exact extension (right, then left; at a node end the neighbour with
// the UNIQUE longest common prefix over the shortest neighbour's length) is replayed on the raw node sequences, 128 characters
// per trip (the reverse strand on a reverse-complemented copy of the reads the row writes first).  A read is MAPPED when a match covers the whole read; more than one such match makes it
// non-unique (MAPQ 0).  HBM-bound byte work: no LDS staging is needed (a read touches <= 2L graph bytes).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"
#include "pg_internal.h"
#include "pg_kmerindex.h"

namespace
{
constexpr uint64_t HASH_B = PG_HASH_B;

struct PathArgs
{
    uint32_t n_reads;
    uint32_t k;
    uint64_t pow_k1;  // HASH_B^(k-1)
    const uint32_t* base_off;
    const char* bases;
    char* bases_rc;  // the reads as the reverse strand sees them, at the same offsets: written by the kernel
    const uint32_t* graph_of_read;
    const PathGraphDev* graphs;
    const KmerEntry* table;
    const uint32_t* filter;  // presence bits of the graphs' k-mers (PathGraphDev::filt_off / filt_mask)
    const uint32_t* pool;
    const uint32_t* node_off;  // raw char offsets per (set-wide) node, n_total + 1
    const char* raw;
    const uint32_t* succ_off;  // per set-wide node
    const uint32_t* succ;
    const uint32_t* pred_off;
    const uint32_t* pred;
    pg_result* results;
    pg_op* ops;
    unsigned long long* ops_counter;
    uint8_t* flags;
    const uint8_t* active;  // nullptr = every read
};

__device__ __forceinline__ uint32_t comp_raw(uint32_t c)
{  // GT!/src/graphutils/SequenceOperations.cpp:66-81 (case-sensitive: anything else becomes 'N')
    switch (c)
    {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}

// Sixteen lanes per read (a 16-lane row of the wavefront = one read, four reads per wavefront).  The stage's time is a chain of
// dependent memory round trips per read, not arithmetic; with ONE thread per read (rounds 1 - 5) a read paid ~25 of them for its
// scan (eight k-mer positions probed per trip) and 20 - 60 more per extension, and a workflow batch of 19 200 reads was 300
// wavefronts on a chip with 1 024 SIMDs: 0.7 ms of pure latency per launch, 1.5 ms beside a fill.  Now
//   * scan: the sixteen lanes hash and probe 128 k-mer positions of a strand in ONE trip (lane i: positions 8 i .. 8 i + 7 of the
//     window, hashes rolled in registers, eight probes in flight per lane); the non-empty slots become a 128-bit candidate mask
//     in LDS that the walk of PathAligner.cpp:92-106 then consumes with bit scans, no further probing;
//   * the reference's character loops (PathOperations.cpp:128-136, 150-159, 202-210, 225-235: runs of equal characters of two
//     byte strings, forwards or backwards) compare 128 characters per trip, eight per lane, the first mismatch found by a ballot;
//   * everything that steers the walk (positions, nodes, lengths) is computed by all sixteen lanes from the same values: control
//     flow is uniform within a row, the four rows of a wavefront diverge from each other.
// Wide loads only where eight characters remain on both sides: nothing is read outside the two strings.
constexpr int PGL = 16;        // lanes per read
constexpr int PWIN = 8 * PGL;  // k-mer positions per scan window

__device__ __forceinline__ uint64_t load8(const char* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

struct Row
{
    int grp, kl;  // the wavefront's row (read) this lane works for, lane within the row
    __device__ uint32_t ballot(bool p) const { return (uint32_t)(__ballot(p) >> (PGL * grp)) & 0xFFFFu; }
    __device__ uint32_t bcast(uint32_t v, int src_lane) const { return (uint32_t)__shfl((int)v, grp * PGL + src_lane); }
};

// number of leading characters x[0..n) and y[0..n) have in common (the same value in every lane of the row)
__device__ __forceinline__ uint32_t common_prefix(const Row& w, const char* x, const char* y, uint32_t n)
{
    for (uint32_t base = 0; base < n; base += (uint32_t)PWIN)
    {
        const uint32_t off = base + 8u * (uint32_t)w.kl;
        uint64_t d = 0;
        if (off + 8 <= n)
            d = load8(x + off) ^ load8(y + off);
        else if (off < n)
            for (uint32_t i = 0; off + i < n; ++i)
                d |= (uint64_t)(uint8_t)(x[off + i] ^ y[off + i]) << (8 * i);
        const uint32_t bal = w.ballot(d != 0);
        if (bal)
        {
            const int fl = __builtin_ctz(bal);
            return w.bcast(off + ((uint32_t)__builtin_ctzll(d | (1ull << 63)) >> 3), fl);
        }
    }
    return n;
}

// number of trailing characters the strings ENDING at xe and ye (exclusive) have in common, at most n
__device__ __forceinline__ uint32_t common_suffix(const Row& w, const char* xe, const char* ye, uint32_t n)
{
    for (uint32_t base = 0; base < n; base += (uint32_t)PWIN)
    {
        const uint32_t off = base + 8u * (uint32_t)w.kl;  // characters between this lane's eight and the end
        uint64_t d = 0;
        if (off + 8 <= n)
            d = load8(xe - off - 8) ^ load8(ye - off - 8);
        else if (off < n)
            for (uint32_t i = 0; off + i < n; ++i)  // (the i-th character from the end goes to the top: clz counts from there)
                d |= (uint64_t)(uint8_t)(xe[-1 - (int)(off + i)] ^ ye[-1 - (int)(off + i)]) << (8 * (7 - i));
        const uint32_t bal = w.ballot(d != 0);
        if (bal)
        {
            const int fl = __builtin_ctz(bal);
            return w.bcast(off + ((uint32_t)__builtin_clzll(d | 1ull) >> 3), fl);
        }
    }
    return n;
}

// A row's view of its graph.  SMALL: the graph's node offsets and successor / predecessor lists sit in the row's LDS block (graphs of
// up to TOPO_N nodes and TOPO_E edges: every site graph the templates make), so the steps of a walk that only ask "how long is this
// node, who follows it" cost an LDS read, not a memory round trip; larger graphs read the tables where they are.
constexpr uint32_t TOPO_N = 32, TOPO_E = 64;
constexpr uint32_t REC_N = 24;  // nodes a walk may add on either side before the record falls back to the two-pass form
struct RowLds
{
    uint64_t hash[PWIN];                  // hashes of the current scan window's k-mers
    uint8_t mask[PGL] __attribute__((aligned(8)));  // ... and which of them the presence filter lets through
    uint32_t node_off[TOPO_N + 1];        // raw character offsets (set-wide values)
    uint32_t succ_off[TOPO_N + 1];        // offsets into succ[] below (graph-local)
    uint32_t pred_off[TOPO_N + 1];
    uint32_t succ[TOPO_E];
    uint32_t pred[TOPO_E];
    uint32_t rec[2][2][REC_N];            // [buffer][left / right][i]: nodes the walks added (closest first)
};

template <bool SMALL> struct Walker
{
    const PathArgs& a;
    const PathGraphDev g;
    const Row w;
    RowLds& t;
    const char* qp;  // the read as this strand sees it (the reverse strand: the copy the row wrote)
    int L;

    __device__ __forceinline__ uint32_t q(int j) const { return (uint8_t)qp[j]; }
    __device__ __forceinline__ uint32_t noff(uint32_t node) const { return SMALL ? t.node_off[node] : a.node_off[g.node_base + node]; }
    __device__ __forceinline__ uint32_t nlen(uint32_t node) const { return noff(node + 1) - noff(node); }
    __device__ __forceinline__ const char* nptr(uint32_t node) const { return a.raw + noff(node); }
    __device__ __forceinline__ uint32_t succ_begin(uint32_t node) const { return SMALL ? t.succ_off[node] : a.succ_off[g.node_base + node]; }
    __device__ __forceinline__ uint32_t succ_at(uint32_t s) const { return SMALL ? t.succ[s] : a.succ[s]; }
    __device__ __forceinline__ uint32_t pred_begin(uint32_t node) const { return SMALL ? t.pred_off[node] : a.pred_off[g.node_base + node]; }
    __device__ __forceinline__ uint32_t pred_at(uint32_t s) const { return SMALL ? t.pred[s] : a.pred[s]; }

    // Extends the seed path `e` anchored at read position qpos (extendPathMatching).  Outputs the final path as
    // (first node, start_pos, last node, end_pos, length, #nodes prepended, #nodes appended) and the new qpos.
    // The prepended / appended node ids go (from the row's first lane) to rec_l[0..] (closest first) / rec_r[0..], at most
    // rec_cap of each (the counts go on).
    __device__ __forceinline__ void extend(const KmerEntry& e, int& qpos, uint32_t& first_node, uint32_t& start_pos, uint32_t& end_pos,
                                           int& length, uint32_t& n_left, uint32_t& n_right, uint32_t* rec_l, uint32_t* rec_r, uint32_t rec_cap) const
    {
        // ---- extendPathEndMatching (PathOperations.cpp:117-189)
        uint32_t node = a.pool[e.pool_off + e.n_nodes - 1];
        uint32_t pos_in_node = e.end_pos + 1;
        int pos_in_query = qpos + (int)a.k;
        n_right = 0;
        bool moved = true;
        while (moved)
        {
            moved = false;
            const uint32_t len = nlen(node);
            if (pos_in_query < L && pos_in_node < len)
            {
                const uint32_t m = common_prefix(w, qp + pos_in_query, nptr(node) + pos_in_node, min((uint32_t)(L - pos_in_query), len - pos_in_node));
                moved = m != 0;
                pos_in_node += m;
                pos_in_query += (int)m;
            }
            if (pos_in_node >= len)
            {
                const uint32_t sb = succ_begin(node), se = succ_begin(node + 1);
                uint32_t min_size = 0xFFFFFFFFu;
                for (uint32_t s = sb; s < se; ++s)
                    min_size = min(min_size, nlen(succ_at(s)));
                uint32_t n_longest = 0, longest = 0, cur = 0;
                for (uint32_t s = sb; s < se; ++s)
                {
                    const uint32_t sn = succ_at(s);
                    const uint32_t p = common_prefix(w, nptr(sn), qp + pos_in_query, min(min_size, (uint32_t)(L - pos_in_query)));
                    if (p > longest)
                    {
                        longest = p;
                        cur = sn;
                        n_longest = 1;
                    }
                    else if (p == longest)
                        ++n_longest;
                }
                if (longest == 0 || n_longest != 1)
                    break;
                if (w.kl == 0 && n_right < rec_cap)
                    rec_r[n_right] = cur;
                ++n_right;
                pos_in_query += (int)longest;
                pos_in_node = longest;
                node = cur;
                moved = true;
            }
        }
        end_pos = pos_in_node - 1;
        const int end_query = pos_in_query;
        // ---- extendPathStartMatching (PathOperations.cpp:191-266)
        node = a.pool[e.pool_off];
        pos_in_node = e.start_pos;
        pos_in_query = qpos;
        n_left = 0;
        moved = true;
        while (moved)
        {
            moved = false;
            if (pos_in_query > 0 && pos_in_node > 0)
            {
                const uint32_t m = common_suffix(w, qp + pos_in_query, nptr(node) + pos_in_node, min((uint32_t)pos_in_query, pos_in_node));
                moved = m != 0;
                pos_in_node -= m;
                pos_in_query -= (int)m;
            }
            if (pos_in_node == 0)
            {
                const uint32_t pb = pred_begin(node), pe = pred_begin(node + 1);
                uint32_t min_size = 0xFFFFFFFFu;
                for (uint32_t s = pb; s < pe; ++s)
                    min_size = min(min_size, nlen(pred_at(s)));
                uint32_t n_longest = 0, longest = 0, cur = 0;
                for (uint32_t s = pb; s < pe; ++s)
                {
                    const uint32_t pn = pred_at(s);
                    const uint32_t ml = common_suffix(w, nptr(pn) + nlen(pn), qp + pos_in_query, min(min_size, (uint32_t)pos_in_query));
                    if (ml > longest)
                    {
                        longest = ml;
                        cur = pn;
                        n_longest = 1;
                    }
                    else if (ml == longest)
                        ++n_longest;
                }
                if (longest == 0 || n_longest != 1)
                    break;
                if (w.kl == 0 && n_left < rec_cap)
                    rec_l[n_left] = cur;
                ++n_left;
                pos_in_query -= (int)longest;
                node = cur;
                pos_in_node = nlen(node) - longest;
                moved = true;
            }
        }
        first_node = node;
        start_pos = pos_in_node;
        qpos = pos_in_query;
        length = end_query - pos_in_query;
    }

    // hash of q[p .. p + k)
    __device__ __forceinline__ uint64_t hash_at(int p) const
    {
        uint64_t h = 0;
        uint32_t c = 0;
        for (; c + 8 <= a.k; c += 8)
        {
            const uint64_t v = load8(qp + p + (int)c);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                h = h * HASH_B + ((v >> (8 * i)) & 0xFFu) + 1;
        }
        for (; c < a.k; ++c)
            h = h * HASH_B + (uint64_t)q(p + (int)c) + 1;
        return h;
    }

    // the word of the graph's presence filter that holds the bit of the k-mer with hash h (a clear bit: the k-mer is in no path)
    __device__ __forceinline__ uint32_t filter_word(uint64_t h) const
    {
        if (h == 0)
            h = 1;
        return a.filter[g.filt_off + (((uint32_t)(h >> 32) & g.filt_mask) >> 5)];
    }
    __device__ __forceinline__ static uint32_t filter_bit(uint64_t h) { return (uint32_t)((h ? h : 1ull) >> 32) & 31u; }

    // pg_kmer_lookup_unique (pg_kmerindex.h) with the k characters compared node by node in runs
    __device__ __forceinline__ bool lookup(uint64_t h, int pos, KmerEntry& out) const
    {
        if (g.tab_mask == 0xFFFFFFFFu)
            return false;
        if (h == 0)
            h = 1;
        uint32_t slot = (uint32_t)(h >> 20) & g.tab_mask;
        for (;;)
        {
            const KmerEntry e = a.table[g.tab_off + slot];  // (the whole entry in one trip: nearly every probe of a candidate is a hit)
            if (e.hash == 0)
                return false;
            if (e.hash == h)
            {
                if (e.count != 1)
                    return false;
                uint32_t ni = 0, p = e.start_pos, c = 0;
                for (;;)
                {
                    const uint32_t node = a.pool[e.pool_off + ni];
                    const uint32_t seg = min(g.k - c, nlen(node) - p);
                    if (common_prefix(w, nptr(node) + p, qp + pos + (int)c, seg) != seg)
                        return false;
                    c += seg;
                    if (c >= g.k)
                        break;
                    if (++ni >= e.n_nodes)
                        return false;
                    p = 0;
                }
                out = e;
                return true;
            }
            slot = (slot + 1) & g.tab_mask;
        }
    }

    // One scan window: the k-mers starting at win .. win + 127 (those that exist: <= last).  Lane i hashes and probes positions
    // win + 8 i .. win + 8 i + 7; hashes go to t.hash[0..127], the "may be a k-mer of the graph" bits (the presence filter's) to t.mask[0..15] (one byte per
    // lane = eight positions).
    __device__ __forceinline__ void scan_window(int win, int last) const
    {
        const int p0 = win + 8 * w.kl;
        const int nb = min(8, last - p0 + 1);  // (<= 0: none of this lane's positions exist)
        uint32_t bits = 0;
        if (nb > 0)
        {
            uint64_t hs[8];
            hs[0] = hash_at(p0);
            uint64_t qo = 0, qi = 0;  // characters leaving / entering the window at p0, p0 + 1, ...
            if (p0 + 8 <= last)
            {
                qo = load8(qp + p0);
                qi = load8(qp + p0 + (int)a.k);
            }
            else
                for (int i = 0; i + 1 < nb; ++i)
                {
                    qo |= (uint64_t)q(p0 + i) << (8 * i);
                    qi |= (uint64_t)q(p0 + (int)a.k + i) << (8 * i);
                }
#pragma unroll
            for (int i = 0; i < 7; ++i)
                hs[i + 1] = (hs[i] - (((qo >> (8 * i)) & 0xFFu) + 1) * a.pow_k1) * HASH_B + ((qi >> (8 * i)) & 0xFFu) + 1;
            uint32_t fw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                fw[i] = filter_word(hs[i < nb ? i : 0]);  // (every load unconditional: eight in flight, not eight round trips)
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                if (i < nb && ((fw[i] >> filter_bit(hs[i])) & 1u))
                    bits |= 1u << i;
                t.hash[8 * w.kl + i] = hs[i];
            }
        }
        t.mask[w.kl] = (uint8_t)bits;
    }
};

// The read as the reverse strand sees it, written at the read's own offset of the second buffer by the row that goes on to read
// it (a kernel of its own for this waited for a free slot beside the fills like any kernel does: 0.6 ms, profiles/r05_e2e_modes.json):
// lane i writes characters 8 i .. 8 i + 7 of every 128.
__device__ __forceinline__ void reverse_complement(const Row& w, const char* __restrict__ src, char* __restrict__ dst, int L)
{
    for (int base = 0; base < L; base += PWIN)
    {
        const int j = base + 8 * w.kl;
        if (j + 8 <= L)
        {
            const uint64_t v = load8(src + L - 8 - j);
            uint64_t o = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o |= (uint64_t)comp_raw((uint32_t)(v >> (8 * (7 - i))) & 0xFFu) << (8 * i);
            __builtin_memcpy(dst + j, &o, 8);
        }
        else
            for (int i = j; i < L; ++i)
                dst[i] = (char)comp_raw((uint8_t)src[L - 1 - i]);
    }
}

// The walk of one read (PathAligner.cpp:75-164) by its row.
template <bool SMALL>
__device__ __forceinline__ void path_row(const PathArgs& a, const Row& w, RowLds& t, uint32_t r, uint32_t off, int L, const PathGraphDev& g)
{
    uint8_t flags = 0;
    int n_full = 0, n_matches = 0;
    const int last = L - (int)a.k;  // the last position a k-mer starts at
    // Some k-mer of the read AS GIVEN (strand 0) is in no path of the graph (the presence filter has no false negatives): the read
    // cannot match any walk of the graph exactly over its whole length, so a gssw fill of that strand scores below L
    // (PG_PATH_FLAG_FWD_ABSENT, read by pg_batch_retire_exact_matches)
    bool fwd_absent = false;
    // the first full-length match, as the walk that found it left it: seed entry, ends, nodes added (in t.rec[best])
    KmerEntry best_e{};
    uint32_t best_sp = 0, best_ep = 0, best_nl = 0, best_nr = 0;
    int first_strand = 0, first_pos = 0;
    uint32_t cur = 0;  // the record buffer the next walk writes
    for (int strand = 0; strand < 2; ++strand)
    {
        Walker<SMALL> wk{ a, g, w, t, (strand == 0 ? a.bases : a.bases_rc) + off, L };
        // PathAligner.cpp:92-106 walks the read one position at a time; nearly every position is a miss (its k-mer is in no path).
        // The window's mask says where the walk has to look at all.
        int pos = 0, win = -PWIN;
        while (pos <= last)
        {
            if (pos >= win + PWIN)
            {
                win = pos;
                wk.scan_window(win, last);
                __threadfence_block();  // (the row reads what its other lanes wrote)
                if (strand == 0)
                {
                    const int existing = min(PWIN, last - win + 1);
                    const int present = __popcll(*(const uint64_t*)&t.mask[0]) + __popcll(*(const uint64_t*)&t.mask[8]);
                    fwd_absent |= present < existing;
                }
            }
            // first candidate at or behind pos
            int at = -1;
            {
                const int rel = pos - win;
                const uint64_t m0 = *(const uint64_t*)&t.mask[0], m1 = *(const uint64_t*)&t.mask[8];
                const uint64_t lo = rel < 64 ? (m0 >> rel) << rel : 0ull;
                const uint64_t hi = rel < 64 ? m1 : (m1 >> (rel - 64)) << (rel - 64);
                if (lo)
                    at = win + __builtin_ctzll(lo);
                else if (hi)
                    at = win + 64 + __builtin_ctzll(hi);
            }
            if (at < 0)
            {
                pos = win + PWIN;
                continue;
            }
            KmerEntry e;
            int next = at + 1;
            if (wk.lookup(t.hash[at - win], at, e))
            {
                int qpos = at;
                uint32_t fn, sp, ep, nl, nr;
                int len;
                wk.extend(e, qpos, fn, sp, ep, len, nl, nr, t.rec[cur][0], t.rec[cur][1], REC_N);
                ++n_matches;
                if (len == L)
                {
                    if (n_full == 0)
                    {
                        first_strand = strand;
                        first_pos = at;
                        best_e = e;
                        best_sp = sp;
                        best_ep = ep;
                        best_nl = nl;
                        best_nr = nr;
                        cur ^= 1u;  // (the later walks write the other buffer)
                    }
                    ++n_full;
                }
                next = qpos + len + 1;  // PathAligner.cpp:104 + the loop increment
            }
            pos = next;
        }
    }
    if (n_matches)
        flags |= 2;
    if (fwd_absent)
        flags |= PG_PATH_FLAG_FWD_ABSENT;
    if (n_full == 0)
    {
        if (w.kl == 0)
            a.flags[r] = flags;
        return;
    }
    // ---- the first full-length match goes out: one element per node ------------------------------------------------------------
    Walker<SMALL> wk{ a, g, w, t, (first_strand == 0 ? a.bases : a.bases_rc) + off, L };
    const KmerEntry e = best_e;
    const uint32_t nl = best_nl, nr = best_nr, sp = best_sp, ep = best_ep;
    const uint32_t n_nodes = nl + e.n_nodes + nr;
    // (a match run longer than an element holds -- PG_OP_MAX_LEN: only reads beyond 4 095 bases have one -- goes out in pieces: at
    // most L / PG_OP_MAX_LEN more elements, reserved up front)
    const uint32_t extra = (uint32_t)L > PG_OP_MAX_LEN ? (uint32_t)L / PG_OP_MAX_LEN : 0u;
    uint32_t base_lo = 0, base_hi = 0;
    if (w.kl == 0)
    {
        const unsigned long long b0 = atomicAdd(a.ops_counter, (unsigned long long)(n_nodes + extra));
        base_lo = (uint32_t)b0;
        base_hi = (uint32_t)(b0 >> 32);
    }
    const unsigned long long base = ((unsigned long long)w.bcast(base_hi, 0) << 32) | w.bcast(base_lo, 0);
    pg_op* ops = a.ops + base;
    // node ids first (into the ops area itself, behind the reserved extra elements, so that the op words written from the front
    // never overtake the ids still to be read), converted to op words below by the row's first lane, which wrote them
    uint32_t* ids = (uint32_t*)ops + extra;
    if (nl > REC_N || nr > REC_N)
    {
        // (a walk over more nodes than the record holds: walked again, recording into the ops area)
        int qpos = first_pos;
        uint32_t fn, sp2, ep2, nl2, nr2;
        int len;
        wk.extend(e, qpos, fn, sp2, ep2, len, nl2, nr2, ids, ids + nl + e.n_nodes, 0xFFFFFFFFu);
        if (w.kl != 0)
            return;
        for (uint32_t i = 0, j = nl ? nl - 1 : 0; i < j; ++i, --j)
        {
            const uint32_t x = ids[i];
            ids[i] = ids[j];
            ids[j] = x;
        }
    }
    else
    {
        if (w.kl != 0)
            return;
        const uint32_t* rl = t.rec[cur ^ 1u][0];
        const uint32_t* rr = t.rec[cur ^ 1u][1];
        for (uint32_t i = 0; i < nl; ++i)  // prepended nodes were produced closest-first
            ids[i] = rl[nl - 1 - i];
        for (uint32_t i = 0; i < nr; ++i)
            ids[nl + e.n_nodes + i] = rr[i];
    }
    uint32_t n_ops = 0;
    for (uint32_t i = 0; i < e.n_nodes; ++i)
        ids[nl + i] = a.pool[e.pool_off + i];
    for (uint32_t i = 0; i < n_nodes; ++i)
    {
        const uint32_t node = ids[i];
        const uint32_t lo = i == 0 ? sp : 0u;
        const uint32_t hi = i == n_nodes - 1 ? ep : wk.nlen(node) - 1;
        uint32_t left = hi - lo + 1;
        do
        {
            const uint32_t piece = left > PG_OP_MAX_LEN ? PG_OP_MAX_LEN : left;
            ops[n_ops++] = PG_OP_MAKE(node, PG_OPC_M, piece);
            left -= piece;
        } while (left);
    }
    pg_result res;
    res.graph_pos = (int32_t)sp;
    res.score = (int16_t)L;
    res.mapq = n_full == 1 ? 60 : 0;
    res.is_unique = n_full == 1 ? 1 : 0;
    res.returned_reverse = first_strand ? 1 : 0;
    res.multi_mask = 0;
    res.n_ops = (uint16_t)n_ops;
    res.ops_off = (uint32_t)base;
    res.strand_score[0] = first_strand ? -1 : (int16_t)L;
    res.strand_score[1] = first_strand ? (int16_t)L : -1;
    res.clipped = 0;
    res.status = PG_STATUS_PATH_ALIGNER;
    a.results[r] = res;
    a.flags[r] = flags | 1;
}

__global__ __launch_bounds__(64) void pg_path_kernel(PathArgs a)
{
    __shared__ RowLds lds[64 / PGL];
    const Row w{ (int)threadIdx.x / PGL, (int)threadIdx.x % PGL };
    RowLds& t = lds[w.grp];
    const uint32_t r = blockIdx.x * (64u / PGL) + (uint32_t)w.grp;
    if (r >= a.n_reads)
        return;
    if (a.active && !a.active[r])
    {
        if (w.kl == 0)
            a.flags[r] = 0;  // (every read's flag is written: no memset in front of the kernel)
        return;
    }
    const uint32_t off = a.base_off[r];
    const int L = (int)(a.base_off[r + 1] - off);
    const PathGraphDev g = a.graphs[a.graph_of_read[r]];
    if (L < (int)a.k || L == 0 || g.tab_mask == 0xFFFFFFFFu)
    {
        if (w.kl == 0)
            a.flags[r] = 0;
        return;
    }
    // the graph's tables into the row's LDS block (their loads are in flight beside the read's)
    const uint32_t sb0 = a.succ_off[g.node_base], pb0 = a.pred_off[g.node_base];
    const uint32_t n_succ = a.succ_off[g.node_base + g.n_nodes] - sb0, n_pred = a.pred_off[g.node_base + g.n_nodes] - pb0;
    const bool small = g.n_nodes <= TOPO_N && n_succ <= TOPO_E && n_pred <= TOPO_E;
    if (small)
    {
        for (uint32_t i = (uint32_t)w.kl; i <= g.n_nodes; i += PGL)
        {
            t.node_off[i] = a.node_off[g.node_base + i];
            t.succ_off[i] = a.succ_off[g.node_base + i] - sb0;
            t.pred_off[i] = a.pred_off[g.node_base + i] - pb0;
        }
        for (uint32_t i = (uint32_t)w.kl; i < n_succ; i += PGL)
            t.succ[i] = a.succ[sb0 + i];
        for (uint32_t i = (uint32_t)w.kl; i < n_pred; i += PGL)
            t.pred[i] = a.pred[pb0 + i];
    }
    reverse_complement(w, a.bases + off, a.bases_rc + off, L);
    __threadfence_block();  // (the row reads what its other lanes wrote: the copy and the tables)
    if (small)
        path_row<true>(a, w, t, r, off, L, g);
    else
        path_row<false>(a, w, t, r, off, L, g);
}

uint64_t hash_str(const char* s, uint32_t k)
{
    uint64_t h = 0;
    for (uint32_t c = 0; c < k; ++c)
        h = h * HASH_B + (uint64_t)(uint8_t)s[c] + 1;
    return h ? h : 1;
}
}  // namespace

void pg_path_index_free(pg_path_index* ix)
{
    if (!ix)
        return;
    if (ix->ev_tables)
    {
        (void)hipEventSynchronize(ix->ev_tables);  // (the staging block goes back below)
        (void)hipEventDestroy(ix->ev_tables);
    }
    if (ix->ev_built)
    {
        if (!ix->build_pending)
            (void)hipEventSynchronize(ix->ev_built);  // (long complete wherever a batch of the set has come back)
        (void)hipEventDestroy(ix->ev_built);
    }
    pg_pinned_put(ix->staging, ix->staging_cap);
    if (ix->d_block)
        (void)pg_dev_free(ix->d_block);  // (d_graphs, d_node_off, d_raw, d_succ_off, d_succ point into it)
    else
    {
        (void)pg_dev_free(ix->d_graphs);
        (void)pg_dev_free(ix->d_node_off);
        (void)pg_dev_free(ix->d_raw);
        (void)pg_dev_free(ix->d_succ_off);
        (void)pg_dev_free(ix->d_succ);
    }
    (void)pg_dev_free(ix->d_table);
    (void)pg_dev_free(ix->d_pool);
    (void)pg_dev_free(ix->d_node_uniq);
    (void)pg_dev_free(ix->d_filter);
    delete ix;
}

template <typename T> static hipError_t up(const std::vector<T>& v, T** d, hipStream_t s)
{
    hipError_t e = pg_dev_alloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(T));
    if (e != hipSuccess || v.empty())
        return e;
    return hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
}

namespace
{
// One length-k path of a graph, as the enumeration meets it: hash of its characters, where it starts and ends, its nodes
// (in occ_pool).  No strings, no maps: a site's graph has ~700 of these, and a std::unordered_map<std::string, {count, path}> cost
// three allocations for each of them -- 100+ us of host time per site, more than everything else `paragraph`'s default cascade
// adds to a batch (profiles/r05_e2e_phases.json).
struct KmerOcc
{
    uint64_t hash;
    uint32_t start, end, pool_off, n_nodes;
};

struct KmerEnumerator
{
    const pg_graphs* G;
    const std::vector<uint32_t>& succ_off;
    const std::vector<uint32_t>& succ;
    uint32_t nb, k;
    std::vector<KmerOcc>& occ;
    std::vector<uint32_t>& occ_pool;
    std::vector<uint32_t> nl;
    // Prefix hashes of every node (P[0] = 0, P[i + 1] = P[i] B + c_i + 1, all mod 2^64) and the powers of B: the hash of the
    // characters [a, b) of a node is P[b] - P[a] B^(b - a), and a k-mer that runs over several nodes is put together from its
    // pieces with one multiplication each -- the same value as hashing its k characters one by one, which is what a third of a
    // site graph's k-mers (the ones that cross a node boundary) used to cost.
    std::vector<uint64_t> pref, pw;
    std::vector<uint32_t> pref_off;

    const char* node_chars(uint32_t node) const { return G->h_seq_raw.data() + G->h_nodeseq_off[nb + node]; }
    // hash of the characters [a, b) of `node`
    uint64_t piece(uint32_t node, uint32_t a, uint32_t b) const
    {
        const uint64_t* P = pref.data() + pref_off[node];
        return P[b] - P[a] * pw[b - a];
    }
    void leaf(uint64_t h, uint32_t start, uint32_t end)
    {
        occ.push_back(KmerOcc{ h ? h : 1, start, end, (uint32_t)occ_pool.size(), (uint32_t)nl.size() });
        occ_pool.insert(occ_pool.end(), nl.begin(), nl.end());
    }
    // depth-first = extendPathEnd (PathOperations.cpp:70-101): `have` characters hashed into h so far, now at `pos` of nl.back()
    void go(uint64_t h, uint32_t have, uint32_t start, uint32_t pos)
    {
        const uint32_t node = nl.back();
        const uint32_t room = G->h_node_len[nb + node] - pos, need = k - have;
        if (need <= room)
        {
            leaf(h * pw[need] + piece(node, pos, pos + need), start, pos + need - 1);
            return;
        }
        h = h * pw[room] + piece(node, pos, pos + room);
        for (uint32_t s = succ_off[nb + node]; s < succ_off[nb + node + 1]; ++s)
        {
            nl.push_back(succ[s]);
            go(h, have + room, start, 0);
            nl.pop_back();
        }
    }
    // every length-k path of the graph in the reference's order (KmerIndex.cpp:76-99: node by node, position by position)
    void run(uint32_t n_nodes)
    {
        pw.assign(k + 1, 1);
        for (uint32_t i = 1; i <= k; ++i)
            pw[i] = pw[i - 1] * HASH_B;
        pref_off.assign(n_nodes, 0);
        size_t total = 0;
        for (uint32_t node = 0; node < n_nodes; ++node)
        {
            pref_off[node] = (uint32_t)total;
            total += G->h_node_len[nb + node] + 1;
        }
        pref.resize(total);
        for (uint32_t node = 0; node < n_nodes; ++node)
        {
            uint64_t* P = pref.data() + pref_off[node];
            const char* s = node_chars(node);
            const uint32_t len = G->h_node_len[nb + node];
            uint64_t h = 0;
            P[0] = 0;
            for (uint32_t c = 0; c < len; ++c)
                P[c + 1] = h = h * HASH_B + (uint64_t)(uint8_t)s[c] + 1;
        }
        for (uint32_t node = 0; node < n_nodes; ++node)
        {
            const uint32_t len = G->h_node_len[nb + node];
            for (uint32_t pos = 0; pos < len; ++pos)
            {
                nl.assign(1, node);
                if (pos + k <= len)
                    leaf(piece(node, pos, pos + k), pos, pos + k - 1);
                else
                    go(0, 0, pos, pos);
            }
        }
    }
};

// the k characters of an occurrence (for telling a repeated sequence from a 64-bit hash collision)
void occ_chars(const pg_graphs* G, uint32_t nb, const KmerOcc& o, const std::vector<uint32_t>& occ_pool, uint32_t k, char* out)
{
    uint32_t got = 0;
    for (uint32_t i = 0; i < o.n_nodes && got < k; ++i)
    {
        const uint32_t node = occ_pool[o.pool_off + i];
        const uint32_t from = i == 0 ? o.start : 0, len = G->h_node_len[nb + node];
        const uint32_t take = std::min(k - got, len - from);
        memcpy(out + got, G->h_seq_raw.data() + G->h_nodeseq_off[nb + node] + from, take);
        got += take;
    }
}

// The graph's table, filled in enumeration order: the first path with a sequence owns the entry (graphtools::KmerIndex keeps
// every path; only the first and the count are ever read), later ones count up.  false = two different sequences with one hash.
bool build_table(const pg_graphs* G, uint32_t nb, uint32_t k, const std::vector<KmerOcc>& occ, const std::vector<uint32_t>& occ_pool,
                 std::vector<KmerEntry>& table, size_t tab_off, uint32_t cap, std::vector<uint32_t>& pool, std::vector<uint32_t>& first_occ)
{
    first_occ.assign(cap, 0xFFFFFFFFu);
    std::vector<char> a(k), b(k);
    for (uint32_t oi = 0; oi < (uint32_t)occ.size(); ++oi)
    {
        const KmerOcc& o = occ[oi];
        uint32_t slot = (uint32_t)(o.hash >> 20) & (cap - 1);
        for (;;)
        {
            KmerEntry& e = table[tab_off + slot];
            if (e.hash == 0)
            {
                e.hash = o.hash;
                e.count = 1;
                e.start_pos = o.start;
                e.end_pos = o.end;
                e.n_nodes = o.n_nodes;
                e.pool_off = (uint32_t)pool.size();
                pool.insert(pool.end(), occ_pool.begin() + o.pool_off, occ_pool.begin() + o.pool_off + o.n_nodes);
                first_occ[slot] = oi;
                break;
            }
            if (e.hash == o.hash)
            {
                occ_chars(G, nb, occ[first_occ[slot]], occ_pool, k, a.data());
                occ_chars(G, nb, o, occ_pool, k, b.data());
                if (memcmp(a.data(), b.data(), k) != 0)
                    return false;
                ++e.count;
                break;
            }
            slot = (slot + 1) & (cap - 1);
        }
    }
    return true;
}

// KmerIndex::Impl::updateKmerCounts (KmerIndex.cpp:118-141): unique k-mers per node / per edge, from the graph's table
void unique_counts(const std::vector<KmerEntry>& table, size_t tab_off, uint32_t cap, const std::vector<uint32_t>& pool, uint32_t n_nodes,
                   std::vector<uint32_t>& node_cnt, std::unordered_map<uint64_t, uint32_t>* edge_cnt)
{
    node_cnt.assign(n_nodes, 0);
    if (edge_cnt)
        edge_cnt->clear();
    for (uint32_t s = 0; s < cap; ++s)
    {
        const KmerEntry& e = table[tab_off + s];
        if (e.hash == 0 || e.count != 1)
            continue;
        for (uint32_t i = 0; i < e.n_nodes; ++i)
        {
            node_cnt[pool[e.pool_off + i]] += 1;
            if (i && edge_cnt)
                (*edge_cnt)[((uint64_t)pool[e.pool_off + i - 1] << 32) | pool[e.pool_off + i]] += 1;
        }
    }
}
}  // namespace

// The host half of pg_build_kmer_index (no device call: tools/ubench/kmer_index_host.cpp times it on a CPU)
const char* pg_build_kmer_index_host(const pg_graphs* G, const std::vector<int32_t>& k_per_graph, PgKmerIndexHost& t, bool want_node_uniq)
{
    std::vector<uint32_t>& succ_off = t.succ_off;
    std::vector<uint32_t>& succ = t.succ;
    std::vector<PathGraphDev>& gd = t.gd;
    std::vector<KmerEntry>& table = t.table;
    std::vector<uint32_t>& pool = t.pool;
    std::vector<uint8_t>& node_uniq = t.node_uniq;
    std::vector<uint32_t>& h_k = t.h_k;
    std::vector<uint32_t>& filter = t.filter;
    filter.clear();
    const uint32_t n_total = (uint32_t)G->h_node_len.size();
    // successor CSR (set-wide node numbering, graph-local ids as values, ascending): counting sort over the predecessor CSR --
    // a node's successors come out ascending because the nodes are visited in ascending order
    succ_off.assign(n_total + 1, 0);
    succ.assign(G->h_pred.size(), 0);
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
        for (uint32_t node = nb; node < ne; ++node)
            for (uint32_t q = G->h_pred_off[node]; q < G->h_pred_off[node + 1]; ++q)
                ++succ_off[nb + G->h_pred[q] + 1];
    }
    for (uint32_t i = 0; i < n_total; ++i)
        succ_off[i + 1] += succ_off[i];
    {
        std::vector<uint32_t> fill(succ_off.begin(), succ_off.end() - 1);
        for (uint32_t g = 0; g < G->n_graphs; ++g)
        {
            const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
            for (uint32_t node = nb; node < ne; ++node)
                for (uint32_t q = G->h_pred_off[node]; q < G->h_pred_off[node + 1]; ++q)
                    succ[fill[nb + G->h_pred[q]]++] = node - nb;
        }
    }
    gd.assign(G->n_graphs, PathGraphDev{});
    table.clear();
    pool.clear();
    node_uniq.assign(n_total, 0);
    h_k.assign(G->n_graphs, 0);
    std::vector<KmerOcc> occ;
    std::vector<uint32_t> occ_pool, first_occ;
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
        std::vector<uint32_t> node_cnt;
        std::unordered_map<uint64_t, uint32_t> edge_cnt;
        const size_t tab_off = table.size(), pool_mark = pool.size();
        uint32_t cap = 0;
        // the table of graph g for k-mers of length kk, appended to `table` / `pool` (dropped again when the length is only tried)
        auto make = [&](uint32_t kk, bool want_edges) -> int {
            table.resize(tab_off);
            pool.resize(pool_mark);
            occ.clear();
            occ_pool.clear();
            KmerEnumerator en{ G, succ_off, succ, nb, kk, occ, occ_pool, {}, {}, {}, {} };
            en.run(ne - nb);
            cap = 4;
            while (cap < 2 * occ.size())
                cap *= 2;
            if (occ.empty())
            {
                cap = 0;
                node_cnt.assign(ne - nb, 0);
                edge_cnt.clear();
                return 0;
            }
            table.resize(tab_off + cap, KmerEntry{});
            if (!build_table(G, nb, kk, occ, occ_pool, table, tab_off, cap, pool, first_occ))
                return -1;
            if (want_node_uniq || want_edges)
                unique_counts(table, tab_off, cap, pool, ne - nb, node_cnt, want_edges ? &edge_cnt : nullptr);
            else
                node_cnt.assign(ne - nb, 0);  // (the path stage never asks which nodes have unique k-mers: that is the KmerFilter's)
            return 0;
        };
        int32_t k = k_per_graph[g];
        if (k > 0)
        {
            if (make((uint32_t)k, false) != 0)
                return "64-bit k-mer hash collision inside one graph";
        }
        else
        {
            // findMinCoveringKmerLength(graph, -k, -k): the first length 10..63 at which every node and every edge is
            // overlapped by at least -k unique k-mers
            const uint32_t need = (uint32_t)(-k);
            k = -1;
            for (int32_t kk = 10; kk < 64 && k < 0; ++kk)
            {
                if (make((uint32_t)kk, true) != 0)
                    return "64-bit k-mer hash collision inside one graph";
                bool any_below = false;
                for (uint32_t node = 0; node < ne - nb && !any_below; ++node)
                {
                    if (node_cnt[node] < need)
                        any_below = true;
                    for (uint32_t s = succ_off[nb + node]; s < succ_off[nb + node + 1] && !any_below; ++s)
                    {
                        auto it = edge_cnt.find(((uint64_t)node << 32) | succ[s]);
                        if ((it == edge_cnt.end() ? 0u : it->second) < need)
                            any_below = true;
                    }
                }
                if (!any_below)
                    k = kk;
            }
            if (k < 0)
                return "no k-mer length in 10..63 covers every node and edge with unique k-mers";
        }
        h_k[g] = (uint32_t)k;
        for (uint32_t node = 0; node < ne - nb; ++node)
            node_uniq[nb + node] = node_cnt[node] > 0;
        gd[g].node_base = nb;
        gd[g].n_nodes = ne - nb;
        gd[g].k = (uint32_t)k;
        gd[g].pow_k1 = 1;
        for (int32_t i = 1; i < k; ++i)
            gd[g].pow_k1 *= HASH_B;
        gd[g].tab_off = tab_off;
        gd[g].tab_mask = cap ? cap - 1 : 0xFFFFFFFFu;
        // presence filter over the table's keys (the table of the length that was kept)
        uint32_t bits = 1024;
        while (bits < 64u * (cap / 2) && bits < (1u << 28))
            bits *= 2;
        gd[g].filt_off = (uint32_t)filter.size();
        gd[g].filt_mask = bits - 1;
        filter.resize(filter.size() + bits / 32, 0u);
        for (uint32_t sidx = 0; sidx < cap; ++sidx)
            if (const uint64_t h = table[tab_off + sidx].hash)
            {
                const uint32_t bit = (uint32_t)(h >> 32) & (bits - 1);
                filter[gd[g].filt_off + (bit >> 5)] |= 1u << (bit & 31u);
            }
    }
    return nullptr;
}

pg_status pg_build_kmer_index(pg_ctx* ctx, pg_graphs* G, const std::vector<int32_t>& k_per_graph, pg_path_index** out, bool want_node_uniq)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // The tables are made in vectors this thread keeps from call to call: a batch's entry table is ~10 MB, and as a fresh vector
    // it cost 58 us per graph in page faults and regrowth alone -- more than enumerating and hashing the k-mers (12 us) and
    // filling the table (8 us) together (tools/ubench/kmer_index_host.cpp).  They are uploaded before this call returns.
    static thread_local PgKmerIndexHost t;
    if (const char* why = pg_build_kmer_index_host(G, k_per_graph, t, want_node_uniq))
        return pg_fail(ctx, PG_ERR_UNSUPPORTED, why);
    std::vector<uint32_t>& succ_off = t.succ_off;
    std::vector<uint32_t>& succ = t.succ;
    std::vector<PathGraphDev>& gd = t.gd;
    std::vector<KmerEntry>& table = t.table;
    std::vector<uint32_t>& pool = t.pool;
    std::vector<uint8_t>& node_uniq = t.node_uniq;
    std::vector<uint32_t>& h_k = t.h_k;
    const std::string& raw = G->h_seq_raw;
    const std::vector<uint32_t>& noff = G->h_nodeseq_off;
    pg_path_index* ix = new pg_path_index();
    ix->h_k = h_k;
    ix->k = 0;
    if (G->n_graphs)
    {
        ix->k = h_k[0];
        for (uint32_t g = 1; g < G->n_graphs; ++g)
            if (h_k[g] != h_k[0])
                ix->k = 0;
    }
    ix->pow_k1 = 1;
    for (uint32_t i = 1; i < ix->k; ++i)
        ix->pow_k1 *= HASH_B;
    std::vector<uint32_t> node_off(noff.begin(), noff.end());
    std::vector<char> rawv(raw.begin(), raw.end());
    hipError_t e = up(gd, &ix->d_graphs, ctx->stream_copy);
    if (e == hipSuccess) e = up(table, &ix->d_table, ctx->stream_copy);
    if (e == hipSuccess) e = up(pool, &ix->d_pool, ctx->stream_copy);
    if (e == hipSuccess) e = up(node_off, &ix->d_node_off, ctx->stream_copy);
    if (e == hipSuccess) e = up(rawv, &ix->d_raw, ctx->stream_copy);
    if (e == hipSuccess) e = up(succ_off, &ix->d_succ_off, ctx->stream_copy);
    if (e == hipSuccess) e = up(succ, &ix->d_succ, ctx->stream_copy);
    if (e == hipSuccess) e = up(node_uniq, &ix->d_node_uniq, ctx->stream_copy);
    if (e == hipSuccess) e = up(t.filter, &ix->d_filter, ctx->stream_copy);
    if (e == hipSuccess) e = pg_stream_wait(ctx->device, ctx->stream_copy);
    if (e != hipSuccess)
    {
        pg_path_index_free(ix);
        return pg_fail(ctx, PG_ERR_HIP, std::string("k-mer index upload: ") + hipGetErrorString(e));
    }
    *out = ix;
    return PG_OK;
}

// ---- the path stage's index made ON THE DEVICE ---------------------------------------------------------------------------------
// Enumerating a site graph's ~700 k-mers, hashing them, filling a 64 KB table and sending it up cost the host 25 - 30 us per site --
// a sixth of what a (site, sample) costs it in `paragraph`'s default cascade.  The host now only COUNTS the length-k paths (a
// dynamic programme over (node, characters left): tens of operations per node) to size the table, the node pool and the presence
// filter; one thread per start position walks the paths (depth-first over the successors in ascending id, as extendPathEnd does),
// hashes them and claims slots with a compare-and-swap.  A k-mer that occurs once -- the only kind a lookup accepts -- has one
// path whoever inserts it; for a repeated one the entry keeps whichever occurrence came first HERE (graphtools::KmerIndex keeps
// the enumeration's first; nothing reads the path of an entry whose count is not 1).  A second pass compares every occurrence
// of a repeated hash with the entry's own characters: two different sequences under one 64-bit hash fail the build, as on the
// host.  The KmerFilter's index (unique-k-mer counts per node and edge, auto-detected lengths) stays with the host builder.
namespace
{
constexpr uint32_t DEV_INDEX_MAX_K = 64;  // depth of a walk's stack (a node has at least one character)

struct IndexBuildArgs
{
    uint32_t n_chars;               // characters of the whole set = start positions
    uint32_t n_total_nodes;
    uint32_t k;
    const uint32_t* node_off;       // raw character offsets per set-wide node (n_total + 1)
    const char* raw;
    const uint32_t* graph_of_node;  // per set-wide node
    const PathGraphDev* graphs;
    const uint32_t* succ_off;
    const uint32_t* succ;
    KmerEntry* table;
    uint32_t* pool;
    uint32_t* pool_next;            // per graph: next free pool slot (starts at the graph's pool base)
    uint32_t* filter;
    uint32_t* error;                // bit0: two sequences with one hash; bit1: internal (table / pool full)
};

// the characters of the path (nodes ids[0..n), starting at `start` of the first) equal those of (ids2, start2)?  k characters.
__device__ bool same_kmer(const IndexBuildArgs& a, uint32_t node_base, uint32_t k, const uint32_t* ids, uint32_t start, const uint32_t* ids2, uint32_t start2)
{
    uint32_t i1 = 0, p1 = start, i2 = 0, p2 = start2;
    uint32_t n1 = node_base + ids[0], n2 = node_base + ids2[0];
    for (uint32_t c = 0; c < k; ++c)
    {
        while (p1 >= a.node_off[n1 + 1] - a.node_off[n1])
        {
            n1 = node_base + ids[++i1];
            p1 = 0;
        }
        while (p2 >= a.node_off[n2 + 1] - a.node_off[n2])
        {
            n2 = node_base + ids2[++i2];
            p2 = 0;
        }
        if (a.raw[a.node_off[n1] + p1] != a.raw[a.node_off[n2] + p2])
            return false;
        ++p1;
        ++p2;
    }
    return true;
}

template <bool VERIFY> __global__ __launch_bounds__(64) void pg_index_build_kernel(IndexBuildArgs a)
{
    const uint32_t c0 = blockIdx.x * 64u + threadIdx.x;
    if (c0 >= a.n_chars)
        return;
    // the node that holds character c0 (binary search over the set's node offsets)
    uint32_t lo = 0, hi = a.n_total_nodes;
    while (lo + 1 < hi)
    {
        const uint32_t mid = (lo + hi) / 2;
        if (a.node_off[mid] <= c0)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t node0 = lo;
    const uint32_t gi = a.graph_of_node[node0];
    const PathGraphDev g = a.graphs[gi];
    if (g.tab_mask == 0xFFFFFFFFu)
        return;
    const uint32_t k = a.k, start = c0 - a.node_off[node0];
    // depth-first walk: level d holds its node (graph-local), the successor it tries next, and the hash / characters before it
    uint32_t ids[DEV_INDEX_MAX_K], next_succ[DEV_INDEX_MAX_K], have_before[DEV_INDEX_MAX_K];
    uint64_t hash_before[DEV_INDEX_MAX_K];
    int depth = 0;
    ids[0] = node0 - g.node_base;
    have_before[0] = 0;
    hash_before[0] = 0;
    next_succ[0] = 0xFFFFFFFFu;  // not entered yet
    while (depth >= 0)
    {
        const uint32_t node = g.node_base + ids[depth];
        const uint32_t len = a.node_off[node + 1] - a.node_off[node];
        const uint32_t pos = depth == 0 ? start : 0u;
        const uint32_t room = len - pos, need = k - have_before[depth];
        if (next_succ[depth] == 0xFFFFFFFFu)
        {
            // first visit: hash this node's share of the k-mer
            const uint32_t take = need <= room ? need : room;
            uint64_t h = hash_before[depth];
            const char* s = a.raw + a.node_off[node] + pos;
            for (uint32_t c = 0; c < take; ++c)
                h = h * HASH_B + (uint64_t)(uint8_t)s[c] + 1;
            if (need <= room)
            {
                // ---- a complete k-mer: nodes ids[0..depth], from `start` of the first to pos + need - 1 of the last
                if (h == 0)
                    h = 1;
                const uint32_t n_nodes = (uint32_t)depth + 1;
                uint32_t slot = (uint32_t)(h >> 20) & g.tab_mask;
                for (uint32_t probes = 0;; ++probes)
                {
                    KmerEntry* e = a.table + g.tab_off + slot;
                    if (!VERIFY)
                    {
                        const unsigned long long old = atomicCAS((unsigned long long*)&e->hash, 0ull, (unsigned long long)h);
                        if (old == 0ull)
                        {
                            const uint32_t at = atomicAdd(&a.pool_next[gi], n_nodes);
                            for (uint32_t i = 0; i < n_nodes; ++i)
                                a.pool[at + i] = ids[i];
                            e->start_pos = start;
                            e->end_pos = pos + need - 1;
                            e->n_nodes = n_nodes;
                            e->pool_off = at;
                            atomicAdd(&e->count, 1u);
                            const uint32_t bit = (uint32_t)(h >> 32) & g.filt_mask;
                            atomicOr(&a.filter[g.filt_off + (bit >> 5)], 1u << (bit & 31u));
                            break;
                        }
                        if (old == (unsigned long long)h)
                        {
                            atomicAdd(&e->count, 1u);
                            break;
                        }
                    }
                    else
                    {
                        const uint64_t eh = e->hash;
                        if (eh == h)
                        {
                            if (e->count > 1u && !same_kmer(a, g.node_base, k, ids, start, a.pool + e->pool_off, e->start_pos))
                                atomicOr(a.error, 1u);
                            break;
                        }
                        if (eh == 0)
                        {
                            atomicOr(a.error, 2u);  // (an occurrence the first pass did not insert)
                            break;
                        }
                    }
                    if (probes > g.tab_mask)
                    {
                        atomicOr(a.error, 2u);
                        break;
                    }
                    slot = (slot + 1) & g.tab_mask;
                }
                --depth;
                continue;
            }
            hash_before[depth + 1] = h;
            have_before[depth + 1] = have_before[depth] + room;
            next_succ[depth] = a.succ_off[node];
        }
        if (next_succ[depth] < a.succ_off[node + 1] && depth + 1 < (int)DEV_INDEX_MAX_K)
        {
            ids[depth + 1] = a.succ[next_succ[depth]++];
            next_succ[depth + 1] = 0xFFFFFFFFu;
            ++depth;
        }
        else
            --depth;
    }
}

// number of length-k paths of graph g and of the node ids they list together (what the table and the pool must hold): f(node, r) =
// paths of r more characters that start at the node's first character
bool count_kmers(const pg_graphs* G, const std::vector<uint32_t>& succ_off, const std::vector<uint32_t>& succ, uint32_t g, uint32_t k, uint64_t& n_occ,
                 uint64_t& n_pool, std::vector<uint64_t>& f, std::vector<uint64_t>& p)
{
    const uint32_t nb = G->h_node_off[g], n = G->h_node_off[g + 1] - nb;
    f.assign((size_t)n * (k + 1), 0);
    p.assign((size_t)n * (k + 1), 0);
    n_occ = n_pool = 0;
    for (uint32_t node = n; node-- > 0;)
    {
        const uint32_t len = G->h_node_len[nb + node];
        if (len == 0)
            return false;  // (a walk's depth is bounded by k only when every node holds a character: the host builder takes the set)
        for (uint32_t r = 1; r <= k; ++r)
        {
            uint64_t paths = 0, ids = 0;
            if (r <= len)
            {
                paths = 1;
                ids = 1;
            }
            else
                for (uint32_t s = succ_off[nb + node]; s < succ_off[nb + node + 1]; ++s)
                {
                    if (succ[s] <= node)
                        return false;  // (not in topological order: the host builder takes the set)
                    const uint64_t fs = f[(size_t)succ[s] * (k + 1) + (r - len)];
                    paths += fs;
                    ids += fs + p[(size_t)succ[s] * (k + 1) + (r - len)];
                }
            f[(size_t)node * (k + 1) + r] = paths;
            p[(size_t)node * (k + 1) + r] = ids;
        }
        // start positions inside the node: the first len - k + 1 (if any) stay inside it, the others run on into the successors
        const uint32_t inside = len >= k ? len - k + 1 : 0;
        n_occ += inside;
        n_pool += inside;
        for (uint32_t pos = inside; pos < len; ++pos)
        {
            const uint32_t rem = len - pos;  // < k
            for (uint32_t s = succ_off[nb + node]; s < succ_off[nb + node + 1]; ++s)
            {
                const uint64_t fs = f[(size_t)succ[s] * (k + 1) + (k - rem)];
                n_occ += fs;
                n_pool += fs + p[(size_t)succ[s] * (k + 1) + (k - rem)];
            }
        }
    }
    return true;
}

// PG_OK with *out == nullptr: this set is not for the device builder (k too long, nodes out of order, sizes beyond 32 bits)
pg_status build_path_index_on_device(pg_ctx* ctx, pg_graphs* G, uint32_t k, pg_path_index** out)
{
    *out = nullptr;
    // The default since the build's two launches moved from the copy stream (where they stood in front of every lane's uploads: 77 - 79 k
    // against 82.5 k sites/s, profiles/r06_path_index_device_ab.jsonl) onto the seed stream of the set's first path stage: the e2e leg
    // with `paragraph`'s default cascade runs as fast either way (80.6 - 81.3 k sites/s) and the host spends 164 - 166 instead of 192 - 194 us
    // per (site, sample) (profiles/r06_path_index_seed_stream_ab.jsonl).  PG_PATH_INDEX_HOST=1: the host builder.
    if (k > DEV_INDEX_MAX_K || getenv("PG_PATH_INDEX_HOST"))
        return PG_OK;
    static thread_local PgKmerIndexHost t;  // (kept from call to call: see pg_build_kmer_index)
    static thread_local std::vector<uint64_t> f, p;
    static thread_local std::vector<uint32_t> graph_of_node, pool_next;
    std::vector<uint32_t>& succ_off = t.succ_off;
    std::vector<uint32_t>& succ = t.succ;
    const uint32_t n_total = (uint32_t)G->h_node_len.size();
    succ_off.assign(n_total + 1, 0);
    succ.assign(G->h_pred.size(), 0);
    graph_of_node.assign(std::max<uint32_t>(n_total, 1), 0);
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
        for (uint32_t node = nb; node < ne; ++node)
        {
            graph_of_node[node] = g;
            for (uint32_t q = G->h_pred_off[node]; q < G->h_pred_off[node + 1]; ++q)
                ++succ_off[nb + G->h_pred[q] + 1];
        }
    }
    for (uint32_t i = 0; i < n_total; ++i)
        succ_off[i + 1] += succ_off[i];
    {
        std::vector<uint32_t> fill(succ_off.begin(), succ_off.end() - 1);
        for (uint32_t g = 0; g < G->n_graphs; ++g)
        {
            const uint32_t nb = G->h_node_off[g], ne = G->h_node_off[g + 1];
            for (uint32_t node = nb; node < ne; ++node)
                for (uint32_t q = G->h_pred_off[node]; q < G->h_pred_off[node + 1]; ++q)
                    succ[fill[nb + G->h_pred[q]]++] = node - nb;
        }
    }
    std::vector<PathGraphDev>& gd = t.gd;
    gd.assign(G->n_graphs, PathGraphDev{});
    pool_next.assign(std::max<uint32_t>(G->n_graphs, 1), 0);
    uint64_t table_entries = 0, pool_entries = 0, filter_words = 0;
    uint64_t pow_k1 = 1;
    for (uint32_t i = 1; i < k; ++i)
        pow_k1 *= HASH_B;
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        uint64_t n_occ = 0, n_pool = 0;
        if (!count_kmers(G, succ_off, succ, g, k, n_occ, n_pool, f, p))
            return PG_OK;
        uint64_t cap = 0;
        if (n_occ)
        {
            cap = 4;
            while (cap < 2 * n_occ)
                cap *= 2;
        }
        uint64_t bits = 1024;
        while (bits < 64u * (cap / 2) && bits < (1u << 28))
            bits *= 2;
        gd[g].node_base = G->h_node_off[g];
        gd[g].n_nodes = G->h_node_off[g + 1] - G->h_node_off[g];
        gd[g].k = k;
        gd[g].pow_k1 = pow_k1;
        gd[g].tab_off = table_entries;
        gd[g].tab_mask = cap ? (uint32_t)(cap - 1) : 0xFFFFFFFFu;
        gd[g].filt_off = (uint32_t)filter_words;
        gd[g].filt_mask = (uint32_t)(bits - 1);
        pool_next[g] = (uint32_t)pool_entries;
        table_entries += cap;
        pool_entries += n_pool;
        filter_words += bits / 32;
        if (cap > (1ull << 31) || pool_entries >= (1ull << 32) || filter_words >= (1ull << 32))
            return PG_OK;
    }
    if (G->h_seq_raw.size() >= (1ull << 32))
        return PG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pg_path_index* ix = new pg_path_index();
    ix->h_k.assign(G->n_graphs, k);
    ix->k = k;
    ix->pow_k1 = pow_k1;
    // one staged copy for the small tables; the table, the pool and the filter are made on the device
    std::vector<uint32_t> node_off(G->h_nodeseq_off.begin(), G->h_nodeseq_off.end());
    uint32_t *d_graph_of_node = nullptr, *d_pool_next = nullptr, *d_error = nullptr;
    void* staging = nullptr;
    size_t staging_cap = 0;
    hipError_t e;
    {
        PgStagedUpload up;
        up.add(gd, &ix->d_graphs);
        up.add(node_off, &ix->d_node_off);
        up.add_raw(G->h_seq_raw.data(), G->h_seq_raw.size(), (void**)&ix->d_raw);
        up.add(succ_off, &ix->d_succ_off);
        up.add(succ, &ix->d_succ);
        up.add(graph_of_node, &d_graph_of_node);
        up.add(pool_next, &d_pool_next);
        e = up.commit_async(ctx->stream_copy, &ix->d_block, &staging, &staging_cap);
    }
    if (e == hipSuccess) e = pg_dev_alloc((void**)&ix->d_table, std::max<uint64_t>(table_entries, 1) * sizeof(KmerEntry));
    if (e == hipSuccess) e = pg_dev_alloc((void**)&ix->d_pool, std::max<uint64_t>(pool_entries, 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = pg_dev_alloc((void**)&ix->d_filter, (filter_words + 1) * sizeof(uint32_t));  // (+ the error word)
    if (e == hipSuccess)
    {
        ix->d_error = ix->d_filter + filter_words;
        ix->d_graph_of_node = d_graph_of_node;
        ix->d_pool_next = d_pool_next;
        ix->build_table_entries = table_entries;
        ix->build_filter_words = filter_words;
        ix->build_n_chars = (uint32_t)G->h_seq_raw.size();
        ix->build_n_total = n_total;
        ix->build_pending = true;
    }
    // Nothing waits here and nothing is launched here: the set's first path stage queues the memsets and the two build launches on
    // its own seed stream behind ev_tables (pg_path_index_ensure_built); the error word comes back with the batch's result sizes
    // (pg_batch_count publishes it) or with its records / flags, and the page-locked block of the upload is handed back when the
    // index is freed.
    ix->staging = staging;
    ix->staging_cap = staging_cap;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_tables, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ix->ev_tables, ctx->stream_copy);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_built, pg_wait_event_flags());
    if (e != hipSuccess)
    {
        (void)hipStreamSynchronize(ctx->stream_copy);  // (the copy may still read the block)
        ix->build_pending = false;
        pg_path_index_free(ix);
        return pg_fail(ctx, PG_ERR_HIP, std::string("k-mer index build: ") + hipGetErrorString(e));
    }
    *out = ix;
    return PG_OK;
}

hipError_t index_build_launch(pg_path_index* ix, hipStream_t stream)
{
    hipError_t e = hipStreamWaitEvent(stream, ix->ev_tables, 0);
    if (e == hipSuccess) e = hipMemsetAsync(ix->d_table, 0, std::max<uint64_t>(ix->build_table_entries, 1) * sizeof(KmerEntry), stream);
    if (e == hipSuccess) e = hipMemsetAsync(ix->d_filter, 0, (ix->build_filter_words + 1) * sizeof(uint32_t), stream);
    if (e != hipSuccess)
        return e;
    IndexBuildArgs a{};
    a.n_chars = ix->build_n_chars;
    a.n_total_nodes = ix->build_n_total;
    a.k = ix->k;
    a.node_off = ix->d_node_off;
    a.raw = ix->d_raw;
    a.graph_of_node = ix->d_graph_of_node;
    a.graphs = ix->d_graphs;
    a.succ_off = ix->d_succ_off;
    a.succ = ix->d_succ;
    a.table = ix->d_table;
    a.pool = ix->d_pool;
    a.pool_next = ix->d_pool_next;
    a.filter = ix->d_filter;
    a.error = ix->d_error;
    if (a.n_chars)
    {
        hipLaunchKernelGGL(pg_index_build_kernel<false>, dim3((a.n_chars + 63) / 64), dim3(64), 0, stream, a);
        hipLaunchKernelGGL(pg_index_build_kernel<true>, dim3((a.n_chars + 63) / 64), dim3(64), 0, stream, a);
        e = hipGetLastError();
    }
    return e;
}
}  // namespace

hipError_t pg_path_index_ensure_built(pg_path_index* ix, hipStream_t stream)
{
    if (!ix || !ix->ev_built)
        return hipSuccess;  // (made on the host: resident since pg_graphs_build_path_index returned)
    if (ix->build_pending)
    {
        hipError_t e = index_build_launch(ix, stream);
        if (e == hipSuccess) e = hipEventRecord(ix->ev_built, stream);
        if (e != hipSuccess)
            return e;
        ix->build_pending = false;
        ix->built_on = stream;
        return hipSuccess;
    }
    return ix->built_on == stream ? hipSuccess : hipStreamWaitEvent(stream, ix->ev_built, 0);
}

const char* pg_path_index_error_text(uint32_t word)
{
    return word & 1u ? "64-bit k-mer hash collision inside one graph (path index)" : "k-mer index build: table or pool overflow (internal)";
}

pg_status pg_path_index_check(pg_ctx* ctx, const pg_graphs* G)
{
    // (the callers' own copies are on the copy stream: the wait is theirs too, index or no index)
    uint32_t word = 0;
    if (G && G->path_index && G->path_index->d_error && !G->path_index->build_pending)  // (never built: no stage has used it)
        HIP_TRY(ctx, hipMemcpyAsync(&word, G->path_index->d_error, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream_copy));
    HIP_TRY(ctx, pg_stream_wait(ctx->device, ctx->stream_copy));
    if (word)
        return pg_fail(ctx, word & 1u ? PG_ERR_UNSUPPORTED : PG_ERR_HIP, pg_path_index_error_text(word));
    return PG_OK;
}

extern "C" pg_status pg_graphs_build_path_index(pg_ctx* ctx, pg_graphs* G, uint32_t kmer_len)
{
    if (!ctx || !G || kmer_len == 0 || kmer_len > 250)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_graphs_build_path_index: bad argument");
    pg_path_index* ix = nullptr;
    pg_status st = build_path_index_on_device(ctx, G, kmer_len, &ix);
    if (st != PG_OK)
        return st;
    if (!ix)
        st = pg_build_kmer_index(ctx, G, std::vector<int32_t>(G->n_graphs, (int32_t)kmer_len), &ix, false);
    if (st != PG_OK)
        return st;
    ix->k = kmer_len;
    ix->pow_k1 = 1;
    for (uint32_t i = 1; i < kmer_len; ++i)
        ix->pow_k1 *= HASH_B;
    pg_path_index_free(G->path_index);
    G->path_index = ix;
    // the extension walks predecessors too: reuse / create the caller-indexed predecessor tables
    if (!G->d_cnt_pred_off)
    {
        HIP_TRY(ctx, up(G->h_pred_off, &G->d_cnt_pred_off, ctx->stream_copy));
        HIP_TRY(ctx, up(G->h_pred, &G->d_cnt_pred, ctx->stream_copy));
        HIP_TRY(ctx, pg_stream_wait(ctx->device, ctx->stream_copy));
    }
    return PG_OK;
}

/* KmerFilter's index (KmerFilter.cpp:52-76): kmer_len > 0 for every graph, or < 0 = auto-detected per graph */
extern "C" pg_status pg_graphs_build_filter_index(pg_ctx* ctx, pg_graphs* G, int32_t kmer_len, uint32_t* kmer_len_of_graph)
{
    if (!ctx || !G || kmer_len == 0 || kmer_len > 250)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_graphs_build_filter_index: bad argument");
    pg_path_index* ix = nullptr;
    const pg_status st = pg_build_kmer_index(ctx, G, std::vector<int32_t>(G->n_graphs, kmer_len), &ix, true);
    if (st != PG_OK)
        return st;
    if (kmer_len_of_graph)
        for (uint32_t g = 0; g < G->n_graphs; ++g)
            kmer_len_of_graph[g] = ix->h_k[g];
    pg_path_index_free(G->filter_index);
    G->filter_index = ix;
    return PG_OK;
}

extern "C" pg_status pg_batch_path_align(pg_ctx* ctx, pg_batch* b)
{
    if (!ctx || !b || !b->graphs)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_path_align: batch not uploaded");
    const pg_graphs* G = b->graphs;
    if (!G->path_index)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_path_align: call pg_graphs_build_path_index first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const pg_path_index* ix = G->path_index;
    b->h_counters_valid = false;
    // The path stage runs on the SEED stream (one priority level up, like the second stream), not behind the fills queued on the
    // main one nor behind the tracebacks that wait for them on the second: it is 0.2 ms of work per batch whose outcome -- after
    // the count pass and the hand-over, which follow it onto this stream -- decides what the batch's fills are.
    const unsigned turn = ctx->seed_streams > 1 ? ctx->seed_turn++ % (unsigned)ctx->seed_streams : 0u;
    if (turn && !ctx->stream_seed_more[turn - 1])
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->stream_seed_more[turn - 1], hipStreamNonBlocking, ctx->side_priority));
    const hipStream_t ps = turn ? ctx->stream_seed_more[turn - 1] : ctx->stream_seed;
    b->seed_stream = ps;
    b->seed_chain = true;
    {
        const pg_status cp = pg_cascade_prepare_early(ctx, b);  // (the hand-over's tables go up beside the kernel below)
        if (cp != PG_OK)
            return cp;
    }
    HIP_TRY(ctx, pg_stage_begin_on(ctx, b, ps));
    HIP_TRY(ctx, pg_path_index_ensure_built(G->path_index, ps));  // (a device-built index is MADE here, by the set's first path stage)
    // (no memsets in front of the kernel: it writes the flag of every read, and the counter is still zero from the upload when this is
    // the batch's first stage -- every dispatch of a seed chain waits for a wavefront slot beside the fills)
    if (!b->ops_counter_fresh)
        HIP_TRY(ctx, hipMemsetAsync(b->d_ops_counter, 0, sizeof(unsigned long long), ps));
    b->ops_counter_fresh = false;
    if (b->cap_bases_rc < b->cap_bases)  // (the reverse strand's copy of the reads: each thread of the kernel writes its own)
    {
        b->park(b->d_bases_rc);  // (a stage call never waits for the batch: pg_internal.h, parked_blocks)
        b->d_bases_rc = nullptr;
        b->cap_bases_rc = 0;
        HIP_TRY(ctx, pg_dev_alloc((void**)&b->d_bases_rc, b->cap_bases));
        b->cap_bases_rc = b->cap_bases;
    }
    PathArgs a{};
    a.n_reads = b->n_reads;
    a.bases_rc = b->d_bases_rc;
    a.k = ix->k;
    a.pow_k1 = ix->pow_k1;
    a.base_off = b->d_base_off;
    a.bases = b->d_bases;
    a.graph_of_read = b->d_graph_of_read;
    a.graphs = ix->d_graphs;
    a.table = ix->d_table;
    a.filter = ix->d_filter;
    a.pool = ix->d_pool;
    a.node_off = ix->d_node_off;
    a.raw = ix->d_raw;
    a.succ_off = ix->d_succ_off;
    a.succ = ix->d_succ;
    a.pred_off = G->d_cnt_pred_off;
    a.pred = G->d_cnt_pred;
    a.results = b->d_results;
    a.ops = b->d_ops;
    a.ops_counter = b->d_ops_counter;
    a.flags = b->d_path_flags;
    a.active = b->has_active ? b->d_active : nullptr;
    if (b->n_reads)
    {
        hipLaunchKernelGGL(pg_path_kernel, dim3((b->n_reads + 3) / 4), dim3(64), 0, ps, a);  // four reads per wavefront, 16 lanes each
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, pg_stage_end_on(ctx, b, ps));
    return PG_OK;
}

extern "C" pg_status pg_batch_download_path_flags(pg_ctx* ctx, pg_batch* b, uint8_t* flags)
{
    if (!ctx || !b || (b->n_reads && !flags))
        return PG_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, pg_batch_wait(ctx, b));
    if (b->n_reads)
        HIP_TRY(ctx, hipMemcpyAsync(flags, b->d_path_flags, b->n_reads, hipMemcpyDeviceToHost, ctx->stream_copy));
    return pg_path_index_check(ctx, b->graphs);  // (waits for the copy stream; a path index built on the device reports here)
}
