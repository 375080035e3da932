// pg_kmer.hip -- k-mer seed stage of the aligner cascade on the device (--kmer-sequence-matching).
//
// Replaces
//   grm::KmerAligner<16>::{setGraph,alignRead}     src/c++/lib/grm/KmerAligner.cpp:120-177, 246-319, 321-538
//   oligo::KmerGenerator / Translator              src/c++/include/oligo/KmerGenerator.hh:56-154, Nucleotides.hh
//
// Host: for every path of every graph the path sequence, its node starts and the (k-mer, position) table sorted
// by (k-mer, position) (windows of k consecutive ACGT/acgt bases, 2 bits per base, first base most significant).
// Device: one WAVEFRONT per read.  Per strand the read's k-mers are sorted in LDS; per (path, strand) the reference's
// merge-join of the sorted read k-mers against the path table (including its quirk: a k-mer that occurs twice in the read
// only joins through its first occurrence) runs lane-parallel and produces candidate diagonals; they are visited in
// ascending offset through a bitmap, scored by a wave-wide Hamming distance over the raw characters and kept in a bounded
// max-heap that replays libstdc++'s push_heap/pop_heap element order (capacity = paths + 2).  CIGARs: the wave writes one
// (node, op) label per read position into LDS; candidates are compared label by label, the best one is run-length encoded.  pickBest: fewest mismatches (<= 2) wins; any equally
// good candidate with a different (CIGAR, position) makes the read BAD_ALIGN.  CIGARs are never materialised as
// strings: two candidates are compared by streaming their run-length elements side by side.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"
#include "pg_internal.h"

namespace
{
constexpr int MAX_PATHS = 126;  // (the candidate heap of paths + 2 entries lives in LDS, sized per launch: KmerArgs::heap_cap)

struct KPathDev
{
    uint32_t seq_off;    // into pathseq[]
    uint32_t len;
    uint32_t start_off;  // into starts[] (pairs: start position, node id)
    uint32_t n_nodes;
    uint32_t kmer_off;   // into kmers[] / kpos[]
    uint32_t n_kmers;
};
struct KGraphDev
{
    uint32_t path_off;
    uint32_t n_paths;
};

struct KmerArgs
{
    uint32_t n_reads;
    uint32_t k;
    const uint32_t* base_off;
    const char* bases;
    const uint32_t* graph_of_read;
    const uint8_t* is_rev;  // BAM strand per read or nullptr
    const KGraphDev* graphs;
    const KPathDev* paths;
    const char* pathseq;
    const uint32_t* starts;
    const uint32_t* kmers;
    const uint32_t* kpos;
    pg_result* results;
    pg_op* ops;
    unsigned long long* ops_counter;
    uint8_t* flags;
    const uint8_t* active;  // nullptr = every read
    // dynamic LDS layout (sized by the batch's longest read and the index's longest path)
    uint32_t n_tab;     // slots of a strand's k-mer table: a power of two, at least 4/3 of the longest read's windows
    uint32_t l_max;     // longest read, padded to a multiple of 4
    uint32_t bm_words;  // candidate-offset bitmap
    uint32_t cache_n;   // path-table entries cached in LDS (paths with more k-mers are searched in global memory)
    uint32_t heap_cap;  // candidate heap entries: the index's largest path count + 2 (a fixed 128 cost 2 of 17 blocks per CU)
};

__device__ __forceinline__ uint32_t comp_raw(uint32_t c)
{
    switch (c)
    {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}
__device__ __forceinline__ uint32_t base2(uint32_t c)
{  // oligo::Translator: ACGT / acgt -> 0..3, everything else invalid (4)
    switch (c)
    {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

struct Cand
{
    uint32_t path;
    uint32_t pos;
    uint32_t reverse;
    uint32_t mm;
};

struct ReadView
{  // both strands of the read as raw characters (forward: as given; reverse: reverseComplement), staged in LDS
    const uint8_t* fwd;
    const uint8_t* rev;
    int L;
    __device__ uint32_t at(int j, bool reverse) const { return reverse ? rev[j] : fwd[j]; }
};

// Per-position labels of one candidate alignment (KmerAligner.cpp:321-472), written by the whole wavefront into `lab`
// (LDS): label = node id << 4 | op for every read position; the graph CIGAR is the run-length encoding of that array.
// Reference positions that are 'N' at the ends of the window are soft-clipped (:326-329); the clip belongs to the first /
// last aligned node.  Returns false (uniformly) when nothing is left to align (no CIGAR at all).
struct CandLabels
{
    int32_t graph_pos;
    int left, right;
    uint32_t matches;
};

__device__ bool label_candidate(
    const KmerArgs& a, const KPathDev& p, const ReadView& rv, bool reverse, uint32_t cand_pos, int lane, uint32_t* lab, CandLabels& out)
{
    const int L = rv.L;
    auto ref = [&](uint32_t i) -> uint32_t { return (uint8_t)a.pathseq[p.seq_off + i]; };
    auto start_of = [&](uint32_t i) -> uint32_t { return a.starts[p.start_off + 2 * i]; };
    auto node_of = [&](uint32_t i) -> uint32_t { return a.starts[p.start_off + 2 * i + 1]; };
    int left = 0;
    while (left < L && ref(cand_pos + left) == 'N')
        ++left;
    int right = 0;
    while (right < L - left && ref(cand_pos + L - 1 - right) == 'N')
        ++right;
    const uint32_t pos = cand_pos + (uint32_t)left;
    uint32_t node_idx = 0;
    for (uint32_t i = 0; i < p.n_nodes; ++i)
        if (start_of(i) <= pos)
            node_idx = i;
    out.graph_pos = (int32_t)(pos - start_of(node_idx));
    out.left = left;
    out.right = right;
    out.matches = 0;
    const int aligned = L - left - right;
    if (aligned <= 0)
        return false;
    // node of the last aligned base (the right clip is printed inside its bracket)
    const uint32_t last_pp = cand_pos + (uint32_t)(L - right - 1);
    uint32_t last_idx = node_idx;
    for (uint32_t i = node_idx; i < p.n_nodes; ++i)
        if (start_of(i) <= last_pp)
            last_idx = i;
    uint32_t m = 0;
    for (int j = lane; j < L; j += 64)
    {
        uint32_t label;
        if (j < left)
            label = (node_of(node_idx) << 4) | PG_OPC_S;
        else if (j >= L - right)
            label = (node_of(last_idx) << 4) | PG_OPC_S | 8u;  // | 8: a right clip never merges with a left one
        else
        {
            const uint32_t pp = cand_pos + (uint32_t)j;
            uint32_t ni = node_idx;
            while (ni + 1 < p.n_nodes && start_of(ni + 1) <= pp)
                ++ni;
            const uint32_t rc = ref(pp), qc = rv.at(j, reverse);
            const uint32_t op = rc == qc ? PG_OPC_M : ((rc == 'N' || qc == 'N') ? PG_OPC_N : PG_OPC_X);
            m += op == PG_OPC_M;
            label = (node_of(ni) << 4) | op;
        }
        lab[j] = label;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1)
        m += __shfl_xor(m, k);
    out.matches = m;
    return true;
}

__device__ void heap_push(Cand* h, int& n, const Cand& v)
{  // std::push_heap with Candidate::lessMismatches (max-heap on mismatches)
    int hole = n;
    ++n;
    int parent = (hole - 1) / 2;
    while (hole > 0 && h[parent].mm < v.mm)
    {
        h[hole] = h[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h[hole] = v;
}

__device__ void heap_pop(Cand* h, int& n)
{  // std::pop_heap + pop_back (libstdc++ __adjust_heap)
    const Cand value = h[n - 1];
    const int len = n - 1;
    int hole = 0, second = 0;
    while (second < (len - 1) / 2)
    {
        second = 2 * (second + 1);
        if (h[second].mm < h[second - 1].mm)
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
    while (hole > 0 && h[parent].mm < value.mm)
    {
        h[hole] = h[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    if (len > 0)
        h[hole] = value;
    n = len;
}

constexpr int KMER_READ_MAX = 512;  // reads beyond 512 bases are left to the later stages

// One WAVEFRONT per read.  The read and its reverse complement are staged in LDS; per strand the 64 lanes put the read's k-mers
// into an LDS table (k-mer value -> first position); each path's sorted table is cached in LDS and joined in parallel: lane j
// takes table entries j, j + 64, ... and -- this is the reference's single-cursor merge (KmerAligner.cpp:246-276) -- only the
// FIRST read occurrence of a k-mer value joins, with every path entry of that value.  Candidate diagonals go into an LDS bitmap, are visited in ascending offset, scored by a wave-wide Hamming
// distance and pushed into the bounded heap by lane 0 (libstdc++ element order).
__global__ __launch_bounds__(64) void pg_kmer_kernel(KmerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kmer_lds[];
    unsigned long long* skey0 = (unsigned long long*)kmer_lds;       // [2][n_tab] (k-mer << 32 | first position), ~0 = empty
    uint32_t* slab0 = (uint32_t*)(skey0 + 2 * a.n_tab);             // [2][l_max] per-position labels (best / rival)
    uint32_t* sbm = slab0 + 2 * a.l_max;                            // [bm_words]
    uint32_t* spk = sbm + a.bm_words;                               // [cache_n] path k-mers
    uint32_t* spp = spk + a.cache_n;                                // [cache_n] path positions
    Cand* sheap = (Cand*)(spp + a.cache_n);                         // [heap_cap]
    int* shn_p = (int*)(sheap + a.heap_cap);
    uint8_t* sread = (uint8_t*)(shn_p + 4);                         // [2][l_max]
    uint32_t* slab[2] = { slab0, slab0 + a.l_max };
    const uint32_t r = blockIdx.x;
    const int lane = (int)threadIdx.x;
    if (r >= a.n_reads || (a.active && !a.active[r]))
        return;
    const uint32_t off = a.base_off[r];
    const int L = (int)(a.base_off[r + 1] - off);
    if (lane == 0)
        a.flags[r] = 0;
    if (L == 0 || L > (int)a.l_max)
        return;
    const KGraphDev g = a.graphs[a.graph_of_read[r]];
    if (g.n_paths == 0 || g.n_paths > MAX_PATHS)
        return;
    for (int j = lane; j < L; j += 64)
    {
        sread[j] = (uint8_t)a.bases[off + j];
        sread[a.l_max + j] = (uint8_t)comp_raw((uint8_t)a.bases[off + L - 1 - j]);
    }
    ReadView rv{ sread, sread + a.l_max, L };
    const int K = (int)a.k;
    const int cap = (int)g.n_paths + 2;
    uint32_t* bm = sbm;
    int& shn = *shn_p;
    if (lane == 0)
        shn = 0;
    __syncthreads();

    const int n_win = L >= K ? L - K + 1 : 0;
    // ---- the read's k-mers (KmerAligner.cpp:120-133), one table per strand: k-mer value -> its FIRST position ---------------
    // The reference sorts (k-mer, position) pairs and walks them beside the path's sorted table with one cursor (:246-276): of
    // the read's occurrences of a k-mer value only the first one in that order -- the smallest position -- meets the path's
    // entries.  That is all the order is used for, so the pairs are not sorted here (a bitonic sort of 2 x 256 keys in LDS was
    // 36 passes, four fifths of the kernel's instructions): each strand's windows go into an open-addressing table keyed by the
    // k-mer value, equal values keep the smaller position (64-bit minimum on value << 32 | position), and the join below takes
    // the table's entries in whatever order they sit -- it only sets bits.
    const int T = (int)a.n_tab;
    const uint32_t tshift = 32u - (uint32_t)__builtin_ctz((uint32_t)T);
    for (int i = lane; i < 2 * T; i += 64)
        skey0[i] = ~0ull;
    __syncthreads();
    {
        // a lane takes a run of consecutive windows: the value rolls on by one base per window
        const int per_lane = (n_win + 63) / 64;
        const int s = lane * per_lane, e = min(s + per_lane, n_win);
        const uint32_t vmask = K < 16 ? (1u << (2 * K)) - 1u : 0xFFFFFFFFu;
        for (int strand = 0; strand < 2 && s < e; ++strand)
        {
            unsigned long long* tab = skey0 + strand * T;
            uint32_t val = 0;
            int run = 0;  // consecutive bases of ACGT ending at the last one taken
            for (int j = s; j < e + K - 1; ++j)
            {
                const uint32_t bb = base2(rv.at(j, strand != 0));
                if (bb <= 3u)
                {
                    val = ((val << 2) | bb) & vmask;
                    ++run;
                }
                else
                {
                    val = 0;
                    run = 0;
                }
                const int i = j - (K - 1);  // the window that ends here
                if (i < s || run < K)
                    continue;
                const unsigned long long key = ((unsigned long long)val << 32) | (uint32_t)i;
                uint32_t slot = (val * 0x9E3779B1u) >> tshift;
                for (;;)
                {
                    unsigned long long cur = tab[slot];
                    if (cur == ~0ull)
                    {
                        cur = atomicCAS(&tab[slot], ~0ull, key);
                        if (cur == ~0ull)
                            break;
                    }
                    if ((uint32_t)(cur >> 32) == val)
                    {
                        atomicMin(&tab[slot], key);
                        break;
                    }
                    slot = (slot + 1u) & (uint32_t)(T - 1);
                }
            }
        }
    }
    __syncthreads();
    // candidates are pushed path by path, forward strand first (KmerAligner.cpp:524-531): the heap's element order depends on it
    for (uint32_t pi = 0; pi < g.n_paths; ++pi)
    {
        const KPathDev p = a.paths[g.path_off + pi];
        if ((int)p.len < L)
            continue;  // every candidate would overhang the path
        const uint32_t n_off = p.len - (uint32_t)L + 1;
        const uint32_t words = (n_off + 31) / 32;
        if (words > a.bm_words)
            continue;  // cannot happen (bm_words is sized by the longest path)
        // the path's sorted (k-mer, position) table: LDS copy when it fits
        const bool cached = p.n_kmers <= a.cache_n;
        if (cached)
            for (uint32_t e = lane; e < p.n_kmers; e += 64)
            {
                spk[e] = a.kmers[p.kmer_off + e];
                spp[e] = a.kpos[p.kmer_off + e];
            }
        const uint32_t* pk = cached ? spk : a.kmers + p.kmer_off;
        const uint32_t* pp = cached ? spp : a.kpos + p.kmer_off;
        for (int strand = 0; strand < 2; ++strand)
        {
            const bool reverse = strand != 0;
            const unsigned long long* skey = skey0 + strand * T;
            for (uint32_t w = lane; w < words; w += 64)
                bm[w] = 0;
            __syncthreads();
            // ---- merge-join ---------------------------------------------------------------------------------------
            for (int j = lane; j < T; j += 64)
            {
                const unsigned long long key = skey[j];
                if (key == ~0ull)
                    continue;
                const uint32_t km = (uint32_t)(key >> 32), sp = (uint32_t)key;  // (sp: the read's first occurrence of this k-mer)
                uint32_t lo = 0, hi = p.n_kmers;
                while (lo < hi)
                {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (pk[mid] < km)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                for (; lo < p.n_kmers && pk[lo] == km; ++lo)
                {
                    const int offset = (int)pp[lo] - (int)sp;
                    if (offset >= 0 && p.len >= (uint32_t)offset + (uint32_t)L)
                        atomicOr(&bm[offset >> 5], 1u << (offset & 31));
                }
            }
            __syncthreads();
            // ---- candidates in ascending offset (std::sort + std::unique), Hamming distance, bounded heap --------------
            for (uint32_t w = 0; w < words; ++w)
            {
                uint32_t bits = bm[w];
                while (bits)
                {
                    const uint32_t bit = (uint32_t)__builtin_ctz(bits);
                    bits &= bits - 1;
                    const uint32_t offset = w * 32 + bit;
                    uint32_t mm = 0;
                    for (int j = lane; j < L; j += 64)
                        mm += rv.at(j, reverse) != (uint8_t)a.pathseq[p.seq_off + offset + (uint32_t)j];
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1)
                        mm += __shfl_xor(mm, m);
                    if (lane == 0)
                    {
                        int hn = shn;
                        heap_push(sheap, hn, Cand{ pi, offset, (uint32_t)reverse, mm });
                        if (hn == cap)
                            heap_pop(sheap, hn);
                        shn = hn;
                    }
                }
            }
            __syncthreads();
        }
    }
    Cand* heap = sheap;
    const int hn = shn;
    if (hn == 0)
        return;
    // ---- pickBest (KmerAligner.cpp:478-516): every lane evaluates the same (uniform) selection
    int bi = 0;
    for (int i = 1; i < hn; ++i)
        if (heap[i].mm < heap[bi].mm)
            bi = i;
    const Cand best = heap[bi];
    if (best.mm > 2)
        return;
    const KPathDev pbest = a.paths[g.path_off + best.path];
    CandLabels lb;
    const bool has_ops = label_candidate(a, pbest, rv, best.reverse != 0, best.pos, lane, slab[0], lb);
    __syncthreads();
    bool bad = false;
    {
        int i = bi + 1;
        while (i < hn)
        {
            int si = i;
            for (int j = i + 1; j < hn; ++j)
                if (heap[j].mm < heap[si].mm)
                    si = j;
            const Cand sb = heap[si];
            if (sb.mm != best.mm)
                break;
            // "different CIGAR or position" (:497-505): the CIGAR strings are equal iff the per-position labels are
            CandLabels l2;
            const bool has2 = label_candidate(a, a.paths[g.path_off + sb.path], rv, sb.reverse != 0, sb.pos, lane, slab[1], l2);
            __syncthreads();
            bool differ = l2.graph_pos != lb.graph_pos || has2 != has_ops;
            if (!differ && has_ops)
            {
                bool d = false;
                for (int j = lane; j < L; j += 64)
                    d = d || slab[0][j] != slab[1][j];
                differ = __any(d);
            }
            __syncthreads();
            if (differ)
            {
                bad = true;
                break;
            }
            i = si + 1;
        }
    }
    // ---- emit the best alignment: run-length encode the labels, all lanes (one lane walking the read twice was a fifth of the
    // kernel): the positions where a run starts go into LDS (the rival's label array is free by now) by ballot + prefix count,
    // element i of the CIGAR is then (label at start i, start i + 1 - start i)
    uint32_t n_ops = 0;
    uint32_t* starts = slab[1];
    __syncthreads();  // (the last comparison's reads of slab[1])
    if (has_ops)
        for (int c0 = 0; c0 < L; c0 += 64)
        {
            const int j = c0 + lane;
            const bool st = j < L && (j == 0 || slab[0][j] != slab[0][j - 1]);
            const unsigned long long m = __ballot(st);
            if (st)
                starts[n_ops + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)j;
            n_ops += (uint32_t)__popcll(m);
        }
    __syncthreads();
    unsigned long long base = 0;
    if (lane == 0)
        base = atomicAdd(a.ops_counter, (unsigned long long)n_ops);
    base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0);
    for (uint32_t i = (uint32_t)lane; i < n_ops; i += 64u)
    {
        const uint32_t s0 = starts[i], s1 = i + 1 < n_ops ? starts[i + 1] : (uint32_t)L;
        const uint32_t cur = slab[0][s0];
        a.ops[base + i] = PG_OP_MAKE(cur >> 4, cur & 7u, s1 - s0);
    }
    if (lane != 0)
        return;
    pg_result res;
    res.graph_pos = lb.graph_pos;
    res.score = (int16_t)lb.matches;
    res.mapq = bad ? 0 : 60;
    res.is_unique = bad ? 0 : 1;
    res.returned_reverse = (uint8_t)best.reverse;
    res.multi_mask = 0;
    res.n_ops = (uint16_t)n_ops;
    res.ops_off = (uint32_t)base;
    res.strand_score[0] = best.reverse ? -1 : (int16_t)lb.matches;
    res.strand_score[1] = best.reverse ? (int16_t)lb.matches : -1;
    res.clipped = (uint16_t)(has_ops ? lb.left + lb.right : 0);
    res.status = PG_STATUS_KMER_ALIGNER;
    a.results[r] = res;
    a.flags[r] = bad ? 4 : 1;  // bit0 MAPPED, bit2 BAD_ALIGN (ambiguous best)
}
}  // namespace

struct pg_kmer_index
{
    uint32_t k = 0;
    uint32_t max_path_len = 0;
    uint32_t max_path_kmers = 0;
    uint32_t max_paths = 0;  // largest path count of a graph of the set
    KGraphDev* d_graphs = nullptr;
    KPathDev* d_paths = nullptr;
    char* d_pathseq = nullptr;
    uint32_t* d_starts = nullptr;
    uint32_t* d_kmers = nullptr;
    uint32_t* d_kpos = nullptr;
};

void pg_kmer_index_free(pg_kmer_index* ix)
{
    if (!ix)
        return;
    (void)pg_dev_free(ix->d_graphs);
    (void)pg_dev_free(ix->d_paths);
    (void)pg_dev_free(ix->d_pathseq);
    (void)pg_dev_free(ix->d_starts);
    (void)pg_dev_free(ix->d_kmers);
    (void)pg_dev_free(ix->d_kpos);
    delete ix;
}

template <typename T> static hipError_t upk(const std::vector<T>& v, T** d, hipStream_t s)
{
    hipError_t e = pg_dev_alloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(T));
    if (e != hipSuccess || v.empty())
        return e;
    return hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
}

extern "C" pg_status pg_graphs_build_kmer_index(
    pg_ctx* ctx, pg_graphs* G, uint32_t kmer_len, const uint32_t* path_off, const uint32_t* path_node_off,
    const uint32_t* path_nodes)
{
    if (!ctx || !G || !path_off || !path_node_off || kmer_len < 2 || kmer_len > 16)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_graphs_build_kmer_index: bad argument (k must be 2..16)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<KGraphDev> gd(G->n_graphs);
    std::vector<KPathDev> pd;
    std::vector<char> pathseq;
    std::vector<uint32_t> starts, kmers, kpos;
    uint32_t max_len = 0, max_kmers = 0;
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nb = G->h_node_off[g], n_nodes = G->h_node_off[g + 1] - nb;
        gd[g].path_off = path_off[g];
        gd[g].n_paths = path_off[g + 1] - path_off[g];
        if (gd[g].n_paths > MAX_PATHS)
            return pg_fail(ctx, PG_ERR_UNSUPPORTED, "more than 126 paths on one graph");
        for (uint32_t p = path_off[g]; p < path_off[g + 1]; ++p)
        {
            KPathDev kp{};
            kp.seq_off = (uint32_t)pathseq.size();
            kp.start_off = (uint32_t)starts.size();
            kp.n_nodes = path_node_off[p + 1] - path_node_off[p];
            uint32_t pos = 0;
            for (uint32_t q = path_node_off[p]; q < path_node_off[p + 1]; ++q)
            {
                const uint32_t node = path_nodes[q];
                if (node >= n_nodes)
                    return pg_fail(ctx, PG_ERR_INVALID, "path node id out of range");
                starts.push_back(pos);
                starts.push_back(node);
                const uint32_t so = G->h_nodeseq_off[nb + node], len = G->h_node_len[nb + node];
                pathseq.insert(pathseq.end(), G->h_seq_raw.begin() + so, G->h_seq_raw.begin() + so + len);
                pos += len;
            }
            kp.len = pos;
            max_len = std::max(max_len, pos);
            // windows of k consecutive valid bases, sorted by (k-mer, position)
            std::vector<std::pair<uint32_t, uint32_t>> ks;
            uint32_t val = 0;
            int run = 0;
            for (uint32_t i = 0; i < pos; ++i)
            {
                const char c = pathseq[kp.seq_off + i];
                int b = (c == 'A' || c == 'a') ? 0 : (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 4;
                if (b > 3)
                {
                    run = 0;
                    val = 0;
                    continue;
                }
                val = (val << 2) | (uint32_t)b;
                if (kmer_len < 16)
                    val &= (1u << (2 * kmer_len)) - 1u;
                if (++run >= (int)kmer_len)
                    ks.emplace_back(val, i - kmer_len + 1);
            }
            std::sort(ks.begin(), ks.end());
            kp.kmer_off = (uint32_t)kmers.size();
            kp.n_kmers = (uint32_t)ks.size();
            max_kmers = std::max(max_kmers, kp.n_kmers);
            for (auto const& e : ks)
            {
                kmers.push_back(e.first);
                kpos.push_back(e.second);
            }
            pd.push_back(kp);
        }
    }
    pg_kmer_index* ix = new pg_kmer_index();
    ix->k = kmer_len;
    ix->max_path_len = max_len;
    ix->max_path_kmers = max_kmers;
    for (auto const& g : gd)
        ix->max_paths = std::max<uint32_t>(ix->max_paths, g.n_paths);
    hipError_t e = upk(gd, &ix->d_graphs, ctx->stream_copy);
    if (e == hipSuccess) e = upk(pd, &ix->d_paths, ctx->stream_copy);
    if (e == hipSuccess) e = upk(pathseq, &ix->d_pathseq, ctx->stream_copy);
    if (e == hipSuccess) e = upk(starts, &ix->d_starts, ctx->stream_copy);
    if (e == hipSuccess) e = upk(kmers, &ix->d_kmers, ctx->stream_copy);
    if (e == hipSuccess) e = upk(kpos, &ix->d_kpos, ctx->stream_copy);
    if (e == hipSuccess) e = pg_stream_wait(ctx->device, ctx->stream_copy);
    if (e != hipSuccess)
    {
        pg_kmer_index_free(ix);
        return pg_fail(ctx, PG_ERR_HIP, std::string("kmer index upload: ") + hipGetErrorString(e));
    }
    pg_kmer_index_free(G->kmer_index);
    G->kmer_index = ix;
    return PG_OK;
}

extern "C" pg_status pg_batch_kmer_align(pg_ctx* ctx, pg_batch* b, uint32_t flags)
{
    if (!ctx || !b || !b->graphs)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_kmer_align: batch not uploaded");
    const pg_graphs* G = b->graphs;
    if (!G->kmer_index)
        return pg_fail(ctx, PG_ERR_INVALID, "pg_batch_kmer_align: call pg_graphs_build_kmer_index first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pg_kmer_index* ix = G->kmer_index;
    const uint32_t words = (ix->max_path_len + 31) / 32 + 1;
    b->h_counters_valid = false;
    b->seed_chain = false;
    {
        const pg_status cp = pg_cascade_prepare_early(ctx, b);  // (a hand-over behind this stage finds its tables on the device)
        if (cp != PG_OK)
            return cp;
    }
    HIP_TRY(ctx, pg_stage_begin(ctx, b));
    if (!(flags & PG_AF_KEEP_RESULTS) || flags == PG_AF_ALL)
        if (!b->ops_counter_fresh)
            HIP_TRY(ctx, hipMemsetAsync(b->d_ops_counter, 0, sizeof(unsigned long long), ctx->stream));
    b->ops_counter_fresh = false;
    KmerArgs a{};
    a.n_reads = b->n_reads;
    a.k = ix->k;
    a.base_off = b->d_base_off;
    a.bases = b->d_bases;
    a.graph_of_read = b->d_graph_of_read;
    a.is_rev = nullptr;
    a.graphs = ix->d_graphs;
    a.paths = ix->d_paths;
    a.pathseq = ix->d_pathseq;
    a.starts = ix->d_starts;
    a.kmers = ix->d_kmers;
    a.kpos = ix->d_kpos;
    a.results = b->d_results;
    a.ops = b->d_ops;
    a.ops_counter = b->d_ops_counter;
    a.flags = b->d_path_flags;
    a.active = b->has_active ? b->d_active : nullptr;
    // reads beyond the stage's 512 bases are left alone (the kernel returns for L > l_max): in the cascade they fall through to
    // the later stages -- one long read must not cost the stage its batch
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < b->n_reads; ++i)
    {
        const uint32_t L = b->h_base_off[i + 1] - b->h_base_off[i];
        if (L <= (uint32_t)KMER_READ_MAX)
            max_len = std::max(max_len, L);
    }
    a.l_max = (max_len + 3u) & ~3u;
    const uint32_t n_win_max = max_len >= ix->k ? max_len - ix->k + 1 : 0;
    a.n_tab = 64;
    while (a.n_tab * 3 < n_win_max * 4)
        a.n_tab <<= 1;
    a.bm_words = words;
    a.cache_n = std::min<uint32_t>(ix->max_path_kmers, 2048u);
    a.heap_cap = (ix->max_paths + 2 + 1u) & ~1u;  // (even: keeps what follows 8-byte aligned whatever sizeof(Cand) is)
    const size_t lds = (size_t)2 * a.n_tab * 8 + (size_t)2 * a.l_max * 4 + (size_t)a.bm_words * 4 + (size_t)2 * a.cache_n * 4
        + (size_t)a.heap_cap * sizeof(Cand) + 16 + (size_t)2 * a.l_max;
    if (b->n_reads)
    {
        if (lds > 48 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)pg_kmer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(pg_kmer_kernel, dim3(b->n_reads), dim3(64), lds, ctx->stream, a);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, pg_stage_end(ctx, b));
    return PG_OK;
}
