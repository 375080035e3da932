// pg_kmer.hip -- k-mer seed stage of the aligner cascade on the device (--kmer-sequence-matching).
//
// Replaces
//   grm::KmerAligner<16>::{setGraph,alignRead}     src/c++/lib/grm/KmerAligner.cpp:120-177, 246-319, 321-538
//   oligo::KmerGenerator / Translator              src/c++/include/oligo/KmerGenerator.hh:56-154, Nucleotides.hh
//
// Host: for every path of every graph the path sequence, its node starts and the (k-mer, position) table sorted
// by (k-mer, position) (windows of k consecutive ACGT/acgt bases, 2 bits per base, first base most significant).
// Device: one thread per read.  Per (path, strand): the reference's merge-join of the read's sorted k-mers against
// the path table (including its quirk: a k-mer that occurs twice in the read only joins through its first
// occurrence) produces candidate diagonals; they are visited in ascending offset through a bitmap, scored by
// Hamming distance over the raw characters and kept in a bounded max-heap that replays libstdc++'s
// push_heap/pop_heap element order (capacity = paths + 2).  pickBest: fewest mismatches (<= 2) wins; any equally
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
constexpr int MAX_PATHS = 30;
constexpr int HEAP_CAP = MAX_PATHS + 2;

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
    uint32_t* bitmap;  // [n_reads][bitmap_words]
    uint32_t bitmap_words;
    pg_result* results;
    pg_op* ops;
    unsigned long long* ops_counter;
    uint8_t* flags;
    const uint8_t* active;  // nullptr = every read
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
{
    const char* bases;
    int L;
    __device__ uint32_t at(int j, bool reverse) const { return reverse ? comp_raw((uint8_t)bases[L - 1 - j]) : (uint8_t)bases[j]; }
};

// Streams the run-length CIGAR elements of one candidate alignment (KmerAligner.cpp:321-472).
struct OpGen
{
    const KmerArgs& a;
    const KPathDev p;
    ReadView rv;
    bool reverse;
    // alignment geometry
    uint32_t pos;  // path position of the first non-clipped base
    int left, right;
    uint32_t node_idx, this_start;
    int length_left, it;
    bool left_pending;
    // per-node state
    int node_remaining;  // bases of the current node still to emit (0 = need a new node)
    uint32_t cur_node;
    bool right_pending;
    int32_t graph_pos;

    __device__ uint32_t ref(uint32_t i) const { return (uint8_t)a.pathseq[p.seq_off + i]; }
    __device__ uint32_t start_of(uint32_t i) const { return a.starts[p.start_off + 2 * i]; }
    __device__ uint32_t node_of(uint32_t i) const { return a.starts[p.start_off + 2 * i + 1]; }

    __device__ void init(uint32_t cand_pos)
    {
        const int L = rv.L;
        left = 0;
        while (left < L && ref(cand_pos + left) == 'N')
            ++left;
        right = 0;
        while (right < L - left && ref(cand_pos + L - 1 - right) == 'N')
            ++right;
        pos = cand_pos + left;
        node_idx = 0;
        for (uint32_t i = 0; i < p.n_nodes; ++i)
            if (start_of(i) <= pos)
                node_idx = i;
        this_start = pos - start_of(node_idx);
        graph_pos = (int32_t)this_start;
        length_left = L - left - right;
        it = left;
        left_pending = left > 0;
        right_pending = false;
        node_remaining = 0;
        cur_node = 0;
    }

    // next element: returns false at the end; (node, op, len)
    __device__ bool next(uint32_t& node, uint32_t& op, uint32_t& len)
    {
        for (;;)
        {
            if (node_remaining > 0)
            {
                if (left_pending)
                {
                    left_pending = false;
                    node = cur_node;
                    op = PG_OPC_S;
                    len = (uint32_t)left;
                    return true;
                }
                // one run of equal ops inside the node
                const uint32_t r0 = this_start + start_of(node_idx);
                auto opat = [&](int j) -> uint32_t {
                    const uint32_t rc = ref(r0 + (uint32_t)j), qc = rv.at(it + j, reverse);
                    return rc == qc ? PG_OPC_M : ((rc == 'N' || qc == 'N') ? PG_OPC_N : PG_OPC_X);
                };
                const uint32_t o = opat(0);
                int run = 1;
                while (run < node_remaining && opat(run) == o)
                    ++run;
                node = cur_node;
                op = o;
                len = (uint32_t)run;
                it += run;
                this_start += (uint32_t)run;
                node_remaining -= run;
                if (node_remaining == 0 && !(right > 0 && length_left == 0))
                {
                    ++node_idx;
                    this_start = 0;
                }
                else if (node_remaining == 0)
                    right_pending = true;
                return true;
            }
            if (right_pending)
            {
                right_pending = false;
                node = cur_node;
                op = PG_OPC_S;
                len = (uint32_t)right;
                ++node_idx;
                this_start = 0;
                return true;
            }
            if (node_idx >= p.n_nodes || length_left <= 0)
                return false;
            int this_length = length_left;
            if (node_idx + 1 < p.n_nodes)
            {
                const int room = (int)(start_of(node_idx + 1) - start_of(node_idx) - this_start);
                this_length = room < length_left ? room : length_left;
            }
            if (this_length > 0)
            {
                cur_node = node_of(node_idx);
                node_remaining = this_length;
                length_left -= this_length;
            }
            else
            {
                ++node_idx;
                this_start = 0;
            }
        }
    }
};

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

__global__ __launch_bounds__(64) void pg_kmer_kernel(KmerArgs a)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= a.n_reads || (a.active && !a.active[r]))
        return;
    const uint32_t off = a.base_off[r];
    const int L = (int)(a.base_off[r + 1] - off);
    a.flags[r] = 0;
    if (L == 0)
        return;
    const KGraphDev g = a.graphs[a.graph_of_read[r]];
    if (g.n_paths == 0 || g.n_paths > MAX_PATHS)
        return;
    ReadView rv{ a.bases + off, L };
    const int K = (int)a.k;
    Cand heap[HEAP_CAP];
    int hn = 0;
    const int cap = (int)g.n_paths + 2;
    uint32_t* bm = a.bitmap + (size_t)r * a.bitmap_words;

    for (uint32_t pi = 0; pi < g.n_paths; ++pi)
    {
        const KPathDev p = a.paths[g.path_off + pi];
        if ((int)p.len < L)
            continue;  // every candidate would overhang the path
        const uint32_t n_off = p.len - (uint32_t)L + 1;
        const uint32_t words = (n_off + 31) / 32;
        for (int strand = 0; strand < 2; ++strand)
        {
            const bool reverse = strand != 0;
            for (uint32_t w = 0; w < words; ++w)
                bm[w] = 0;
            // merge-join (KmerAligner.cpp:246-276).  The read's k-mers must be visited in (k-mer, position) order
            // with ONE forward cursor over the path table: process the read's k-mers in sorted order by repeatedly
            // extracting the next smallest (k-mer, position) -- O(n^2) over <= L windows, no per-thread arrays.
            uint32_t cursor = 0;
            uint64_t last_key = 0;
            bool have_last = false;
            for (;;)
            {
                // next smallest (kmer, pos) strictly greater than last_key
                uint64_t best_key = ~0ull;
                uint32_t val = 0;
                int run = 0;
                for (int i = 0; i < L; ++i)
                {
                    const uint32_t b = base2(rv.at(i, reverse));
                    if (b > 3)
                    {
                        run = 0;
                        val = 0;
                        continue;
                    }
                    val = (val << 2) | b;
                    if (K < 16)
                        val &= (1u << (2 * K)) - 1u;
                    ++run;
                    if (run >= K)
                    {
                        const uint64_t key = ((uint64_t)val << 32) | (uint32_t)(i - K + 1);
                        if ((!have_last || key > last_key) && key < best_key)
                            best_key = key;
                    }
                }
                if (best_key == ~0ull)
                    break;
                last_key = best_key;
                have_last = true;
                const uint32_t km = (uint32_t)(best_key >> 32), sp = (uint32_t)best_key;
                while (cursor < p.n_kmers && a.kmers[p.kmer_off + cursor] < km)
                    ++cursor;
                while (cursor < p.n_kmers && a.kmers[p.kmer_off + cursor] == km)
                {
                    const int offset = (int)a.kpos[p.kmer_off + cursor] - (int)sp;
                    if (offset >= 0 && p.len >= (uint32_t)offset + (uint32_t)L)
                        bm[offset >> 5] |= 1u << (offset & 31);
                    ++cursor;
                }
            }
            // candidates in ascending offset (std::sort + std::unique), Hamming distance, bounded heap
            for (uint32_t w = 0; w < words; ++w)
            {
                uint32_t bits = bm[w];
                while (bits)
                {
                    const uint32_t bit = (uint32_t)__builtin_ctz(bits);
                    bits &= bits - 1;
                    const uint32_t offset = w * 32 + bit;
                    uint32_t mm = 0;
                    for (int j = 0; j < L; ++j)
                        mm += rv.at(j, reverse) != (uint8_t)a.pathseq[p.seq_off + offset + (uint32_t)j];
                    heap_push(heap, hn, Cand{ pi, offset, (uint32_t)reverse, mm });
                    if (hn == cap)
                        heap_pop(heap, hn);
                }
            }
        }
    }
    if (hn == 0)
        return;
    // ---- pickBest (KmerAligner.cpp:478-516)
    int bi = 0;
    for (int i = 1; i < hn; ++i)
        if (heap[i].mm < heap[bi].mm)
            bi = i;
    const Cand best = heap[bi];
    if (best.mm > 2)
        return;
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
            OpGen g1{ a, a.paths[g.path_off + best.path], rv, best.reverse != 0 };
            OpGen g2{ a, a.paths[g.path_off + sb.path], rv, sb.reverse != 0 };
            g1.init(best.pos);
            g2.init(sb.pos);
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
    OpGen gen{ a, a.paths[g.path_off + best.path], rv, best.reverse != 0 };
    gen.init(best.pos);
    uint32_t n_ops = 0, score = 0, clipped = 0;
    {
        uint32_t nd, op, len;
        while (gen.next(nd, op, len))
            ++n_ops;
    }
    const unsigned long long base = atomicAdd(a.ops_counter, (unsigned long long)n_ops);
    gen.init(best.pos);
    {
        uint32_t nd, op, len, e = 0;
        while (gen.next(nd, op, len))
        {
            a.ops[base + e++] = (nd << 20) | (op << 16) | (len & 0xFFFFu);
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
    res.returned_reverse = (uint8_t)best.reverse;
    res.multi_mask = 0;
    res.n_ops = (uint16_t)n_ops;
    res.ops_off = (uint32_t)base;
    res.strand_score[0] = best.reverse ? -1 : (int16_t)score;
    res.strand_score[1] = best.reverse ? (int16_t)score : -1;
    res.clipped = (uint16_t)clipped;
    res.status = PG_STATUS_KMER_ALIGNER;
    a.results[r] = res;
    a.flags[r] = bad ? 4 : 1;  // bit0 MAPPED, bit2 BAD_ALIGN (ambiguous best)
}
}  // namespace

struct pg_kmer_index
{
    uint32_t k = 0;
    uint32_t max_path_len = 0;
    KGraphDev* d_graphs = nullptr;
    KPathDev* d_paths = nullptr;
    char* d_pathseq = nullptr;
    uint32_t* d_starts = nullptr;
    uint32_t* d_kmers = nullptr;
    uint32_t* d_kpos = nullptr;
    uint32_t* d_bitmap = nullptr;
    size_t bitmap_cap = 0;
};

void pg_kmer_index_free(pg_kmer_index* ix)
{
    if (!ix)
        return;
    (void)hipFree(ix->d_graphs);
    (void)hipFree(ix->d_paths);
    (void)hipFree(ix->d_pathseq);
    (void)hipFree(ix->d_starts);
    (void)hipFree(ix->d_kmers);
    (void)hipFree(ix->d_kpos);
    (void)hipFree(ix->d_bitmap);
    delete ix;
}

template <typename T> static hipError_t upk(const std::vector<T>& v, T** d, hipStream_t s)
{
    hipError_t e = hipMalloc((void**)d, std::max<size_t>(v.size(), 1) * sizeof(T));
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
    uint32_t max_len = 0;
    for (uint32_t g = 0; g < G->n_graphs; ++g)
    {
        const uint32_t nb = G->h_node_off[g], n_nodes = G->h_node_off[g + 1] - nb;
        gd[g].path_off = path_off[g];
        gd[g].n_paths = path_off[g + 1] - path_off[g];
        if (gd[g].n_paths > MAX_PATHS)
            return pg_fail(ctx, PG_ERR_UNSUPPORTED, "more than 30 paths on one graph");
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
    hipError_t e = upk(gd, &ix->d_graphs, ctx->stream);
    if (e == hipSuccess) e = upk(pd, &ix->d_paths, ctx->stream);
    if (e == hipSuccess) e = upk(pathseq, &ix->d_pathseq, ctx->stream);
    if (e == hipSuccess) e = upk(starts, &ix->d_starts, ctx->stream);
    if (e == hipSuccess) e = upk(kmers, &ix->d_kmers, ctx->stream);
    if (e == hipSuccess) e = upk(kpos, &ix->d_kpos, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
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
    const size_t need = (size_t)std::max<uint32_t>(b->n_reads, 1) * words;
    if (need > ix->bitmap_cap)
    {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ix->d_bitmap);
        ix->d_bitmap = nullptr;
        ix->bitmap_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ix->d_bitmap, need * sizeof(uint32_t)));
        ix->bitmap_cap = need;
    }
    HIP_TRY(ctx, pg_stage_begin(ctx, b));
    if (!(flags & PG_AF_KEEP_RESULTS) || flags == PG_AF_ALL)
        HIP_TRY(ctx, hipMemsetAsync(b->d_ops_counter, 0, sizeof(unsigned long long), ctx->stream));
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
    a.bitmap = ix->d_bitmap;
    a.bitmap_words = words;
    a.results = b->d_results;
    a.ops = b->d_ops;
    a.ops_counter = b->d_ops_counter;
    a.flags = b->d_path_flags;
    a.active = b->has_active ? b->d_active : nullptr;
    if (b->n_reads)
    {
        hipLaunchKernelGGL(pg_kmer_kernel, dim3((b->n_reads + 63) / 64), dim3(64), 0, ctx->stream, a);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, pg_stage_end(ctx, b));
    return PG_OK;
}
