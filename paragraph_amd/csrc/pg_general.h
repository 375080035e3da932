// pg_general.h -- the GENERAL form of the gssw stage: what the packed wavefront kernels (pg_fill.hip / pg_trace.hip) cannot
// hold -- reads longer than PG_MAX_READ_LEN (f16 carries exact integers only below 2048) and graphs longer than 65 519
// columns (16-bit column fields of the per-node keys) -- still goes through the device, slowly: one THREAD per fill, rows
// looped instead of register-resident, 32-bit column and cell indices, scores as plain 16-bit integers.  The reference has no
// such bounds (external/gssw/gssw.c:527-786, src/c++/lib/grm/GraphAligner.cpp:110-167); a run over thousands of sites must
// not lose a site to one long read.
//
// Same arithmetic as SURVEY.md 8(a'): match +1, mismatch -4, code 4 against anything 0, gap open 6, extend 1, unsigned
// saturating subtraction; per node the seeds are the lane-wise maxima over the predecessors' last columns
// (gssw_create_seed_byte); max_node = first node with the strictly largest score (gssw.c:4015); alignsEndAtMultNodes incl.
// its byte-pointer view of a word-mode matrix (GraphAligner.cpp:170-212).  The traceback follows pg_trace.hip decision by
// decision (E / F re-derived from H), one cell at a time.
//
// Everything here is plain scalar code callable from a kernel thread and from the host (tests/host_cpp/test_general.cpp runs
// it on the CPU against the reference's gssw.c before it ever meets a GPU); the product only calls it from
// pg_general.hip's kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/paragraph_amd.h"
#include "pg_device.h"

#define PG_GEN_MAX_READ_LEN 16000  // scores, CIGAR element lengths and n_ops stay inside their 16-bit fields
#define PG_GEN_NODE_BYTES 16       // per fill and node: the node's key and first-half maximum (PgGenRead::node_off)

// Workspace of one read on the general path (byte offsets from the start of the general workspace).
struct PgGenRead
{
    uint32_t read;   // index in the batch
    uint32_t graph;
    uint64_t h_off;     // int16 H[strand][column][row] of the two forward-graph fills
    uint64_t seed_off;  // int16 [fill 0..3][2 (H of the last column, E of the next column)][node][row]
    uint64_t col_off;   // int16 [fill][2 (H of the previous column, E of this column)][row]
    uint64_t node_off;  // [fill][node] 16 bytes: u64 key of the node's first maximum (score | inverted column | inverted row),
                        // u32 maximum over the first half of its cells, u32 unused (the scalar fill of the CPU test keeps two
                        // int32 there: maximum, first-half maximum)
    uint64_t ops_off;   // uint32 CIGAR scratch, pg_gen_ops_cap(L) elements
};

// kernel arguments of the general stage (pg_general.hip)
struct PgGenArgs
{
    const PgGenRead* reads;
    uint32_t n;
    uint32_t flags;    // PG_AF_*
    uint32_t max_len;  // longest read of the launch (sizes the fill kernel's LDS columns)
    const PgGraphDev* graphs;
    const PgNode* nodes;
    const uint32_t* preds;
    const char* seqchars;
    const uint32_t* base_off;
    const char* bases;
    uint8_t* ws;          // general workspace (PgGenRead offsets)
    PgFillSummary* fsum;  // [read of the launch][dir * 2 + strand]
    pg_result* results;   // [read of the batch]
    pg_op* ops;
    unsigned long long* ops_counter;
};
hipError_t pg_launch_general(const PgGenArgs& args, hipStream_t stream);

static inline __host__ __device__ uint32_t pg_gen_ops_cap(uint32_t L) { return L + 40u; }  // + pieces of runs beyond PG_OP_MAX_LEN
static inline __host__ __device__ uint64_t pg_gen_align8(uint64_t x) { return (x + 7u) & ~(uint64_t)7u; }
// bytes of workspace one read needs on a graph with `ncols` columns and `n_nodes` nodes
static inline __host__ __device__ uint64_t pg_gen_read_bytes(uint64_t L, uint64_t ncols, uint64_t n_nodes)
{
    return pg_gen_align8(2 * ncols * L * 2) + pg_gen_align8(4 * 2 * n_nodes * L * 2) + pg_gen_align8(4 * 2 * L * 2)
        + pg_gen_align8(4 * n_nodes * PG_GEN_NODE_BYTES) + pg_gen_align8((uint64_t)pg_gen_ops_cap((uint32_t)L) * 4);
}

namespace pggen
{
static inline __host__ __device__ uint32_t nt_code(uint32_t c)
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
static inline __host__ __device__ uint32_t upper_c(uint32_t c) { return (c >= 'a' && c <= 'z') ? c - 32u : c; }
static inline __host__ __device__ uint32_t comp_c(uint32_t c)
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
static inline __host__ __device__ int sub_score(uint32_t a, uint32_t b) { return (a == 4u || b == 4u) ? 0 : (a == b ? 1 : -4); }
static inline __host__ __device__ int sat_sub(int a, int b) { return a > b ? a - b : 0; }
static inline __host__ __device__ int imax(int a, int b) { return a > b ? a : b; }

// The string a fill aligns (GraphAligner.cpp:315-337): forward graph: toUpper(bases) / reverseComplement(bases); reversed
// graph: the same two strings reversed.
static inline __host__ __device__ uint32_t query_char(const char* bases, int L, int dir, int strand, int j)
{
    const uint32_t f = (uint8_t)bases[j], r = (uint8_t)bases[L - 1 - j];
    if (strand == 0)
        return upper_c(dir == 0 ? f : r);
    return comp_c(dir == 0 ? r : f);
}

// Upper-cased character of column i of node `id` in graph direction `dir` (the reversed graph: node n-1-id, sequence reversed).
static inline __host__ __device__ uint32_t ref_char(
    const PgGraphDev& gd, const PgNode* nodes_all, const char* seqchars, int dir, uint32_t id, uint32_t len, uint32_t i)
{
    const PgNode* fwd = nodes_all + gd.dir[0].node_off;
    if (dir == 0)
        return (uint8_t)seqchars[gd.seq_off + fwd[id].col_start + i];
    return (uint8_t)seqchars[gd.seq_off + fwd[gd.dir[0].n_nodes - 1 - id].col_start + (len - 1 - i)];
}

// One fill = one (read, strand, graph direction): gssw_graph_fill (gssw.c:3897-4044) in its textbook form, column by column
// with two rolling columns; H of every cell goes to `H` when it is given (forward graph: the traceback reads it).
static inline __host__ __device__ void fill(
    const PgGraphDev& gd, const PgNode* nodes_all, const uint32_t* preds, const char* seqchars, int dir, int strand,
    const char* bases, int L, int16_t* H, int16_t* seedH, int16_t* seedE, int16_t* colH, int16_t* colE, int32_t* node_max,
    PgFillSummary* out)
{
    const PgNode* nodes = nodes_all + gd.dir[dir].node_off;
    const uint32_t n_nodes = gd.dir[dir].n_nodes;
    int best = 0, best_node = 0, best_i = -1, best_j = 0;
    for (uint32_t id = 0; id < n_nodes; ++id)
    {
        const PgNode nd = nodes[id];
        // seeds: lane-wise maxima over the predecessors (ascending ids) of their last column's H and next-column E
        for (int j = 0; j < L; ++j)
        {
            int sh = 0, se = 0;
            for (uint32_t p = 0; p < nd.n_pred; ++p)
            {
                const size_t at = (size_t)preds[nd.pred_off + p] * (size_t)L + (size_t)j;
                sh = imax(sh, seedH[at]);
                se = imax(se, seedE[at]);
            }
            colH[j] = (int16_t)sh;
            colE[j] = (int16_t)se;
        }
        int nb = 0, nb_i = -1, nb_j = 0, first_half = 0;
        const uint64_t half = ((uint64_t)nd.len * (uint64_t)L + 1u) / 2u;
        for (uint32_t i = 0; i < nd.len; ++i)
        {
            const uint32_t rcode = nt_code(ref_char(gd, nodes_all, seqchars, dir, id, nd.len, i));
            int16_t* Hc = H ? H + ((size_t)nd.col_start + i) * (size_t)L : nullptr;
            int diag = 0, hleft = 0, f = 0;
            for (int j = 0; j < L; ++j)
            {
                const int hp = colH[j];  // H of the previous column (or the seed)
                const int e = colE[j];   // E of this column
                f = j > 0 ? imax(sat_sub(f, PG_GAP_EXT), sat_sub(hleft, PG_GAP_OPEN)) : 0;
                const int d = j > 0 ? diag : 0;
                int h = imax(d + sub_score(rcode, nt_code(query_char(bases, L, dir, strand, j))), 0);
                h = imax(h, imax(e, f));
                diag = hp;
                hleft = h;
                colH[j] = (int16_t)h;
                colE[j] = (int16_t)imax(sat_sub(e, PG_GAP_EXT), sat_sub(h, PG_GAP_OPEN));  // E of the next column
                if (Hc)
                    Hc[j] = (int16_t)h;
                if (h > nb)
                {  // first cell in (column, row) order holding the node's maximum: smallest column, in it the smallest row
                    nb = h;
                    nb_i = (int)i;
                    nb_j = j;
                }
                if ((uint64_t)i * (uint64_t)L + (uint64_t)j < half && h > first_half)
                    first_half = h;
            }
        }
        for (int j = 0; j < L; ++j)
        {
            seedH[(size_t)id * (size_t)L + (size_t)j] = colH[j];
            seedE[(size_t)id * (size_t)L + (size_t)j] = colE[j];
        }
        node_max[2 * id] = nb;
        node_max[2 * id + 1] = first_half;
        if (nb > best)
        {
            best = nb;
            best_node = (int)id;
            best_i = nb_i;
            best_j = nb_j;
        }
    }
    // alignsEndAtMultNodes (GraphAligner.cpp:170-212): the top score ends in more than one node.  A fill whose top score
    // reaches 251 was redone by gssw in 16-bit words (gssw.c:380, 4100-4104), and the scan then reads len * readLen BYTES of
    // that matrix: only the low bytes of the first half of a node's cells can hold a top score, which must be <= 255.
    uint32_t hits = 0;
    for (uint32_t id = 0; id < n_nodes; ++id)
    {
        if (best < 251)
            hits += node_max[2 * id] == best;
        else if (best <= 255)
            hits += node_max[2 * id + 1] == best;
    }
    out->score = best;
    out->max_node = best_node;
    out->ref_end = best > 0 ? best_i : -1;
    out->read_end = best > 0 ? best_j : 0;
    out->end_col = best > 0 ? (int32_t)(nodes[best_node].col_start + (uint32_t)best_i) : -1;
    out->multi = hits > 1 ? 1 : 0;
    out->pad[0] = out->pad[1] = 0;
}

struct Emitter
{
    uint32_t* slot;  // filled from the tail backwards (the walk runs from the end of the alignment to its start)
    uint32_t cap, n;
    uint32_t last_node, last_op, last_len;
    bool overflow, node_has_ops;

    __host__ __device__ void flush()
    {
        if (last_op == 0xFFu)
            return;
        // a run longer than an element holds goes out as several elements of the same node and op (only reads beyond 4 095 bases)
        uint32_t left = last_len;
        do
        {
            const uint32_t piece = left > PG_OP_MAX_LEN ? PG_OP_MAX_LEN : left;
            if (n < cap)
                slot[cap - 1 - n++] = PG_OP_MAKE(last_node, last_op, piece);
            else
                overflow = true;
            left -= piece;
        } while (left);
    }
    __host__ __device__ void emit(uint32_t node, uint32_t op, uint32_t len)
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

// Strand pick (GraphAligner.cpp:340-401) + graph traceback (gssw.c:2621-3537, 1112-1818) of ONE read from the four fill
// summaries and the H / seed arrays of the chosen forward-graph fill.  Writes *res and, tail-aligned, the read's CIGAR
// elements into `scratch` (the caller copies them out); returns the number of elements.
static inline __host__ __device__ uint32_t pick_and_trace(
    const PgGraphDev& gd, const PgNode* nodes_all, const uint32_t* preds, const char* seqchars, const char* bases, int L,
    uint32_t flags, const PgFillSummary* fsum /* [dir * 2 + strand] */, const int16_t* Hboth /* [strand][col][row] */,
    const int16_t* seeds /* [fill][2][node][row] */, uint32_t* scratch, pg_result* res_out)
{
    const bool both = (flags & PG_AF_BOTH_STRANDS) != 0;
    const bool revg = (flags & PG_AF_REVERSE_GRAPH) != 0;
    const int m0 = fsum[0].multi;
    const int m1 = both ? fsum[1].multi : 0;
    const int m2 = revg ? fsum[2].multi : 0;
    const int m3 = (revg && both) ? fsum[3].multi : 0;
    const bool fwd_unique = !m0 && !m2;
    const bool rev_unique = !m1 && !m3;
    bool return_reverse = false;
    if (!fwd_unique && rev_unique && both)
        return_reverse = true;
    else if (fwd_unique && !rev_unique)
        return_reverse = false;
    else if (both)
        return_reverse = fsum[0].score < fsum[1].score;
    const int s = return_reverse ? 1 : 0;
    const bool unique = return_reverse ? rev_unique : fwd_unique;
    const PgFillSummary fs = fsum[s];

    pg_result res;
    res.graph_pos = 0;
    res.score = (int16_t)fs.score;
    res.mapq = unique ? 60 : 0;
    res.is_unique = unique ? 1 : 0;
    res.returned_reverse = return_reverse ? 1 : 0;
    res.multi_mask = (uint8_t)(m0 | (m1 << 1) | (m2 << 2) | (m3 << 3));
    res.n_ops = 0;
    res.ops_off = 0;
    res.strand_score[0] = (int16_t)fsum[0].score;
    res.strand_score[1] = (int16_t)(both ? fsum[1].score : -1);
    res.clipped = 0;
    res.status = 0;
    if (fs.score <= 0)
    {
        res.status = 1;  // all-zero fill: empty CIGAR at position 0 (gssw.c:2728-2732, 2778)
        *res_out = res;
        return 0;
    }

    const PgNode* nodes = nodes_all + gd.dir[0].node_off;
    const uint32_t n_nodes = gd.dir[0].n_nodes;
    const char* refc = seqchars + gd.seq_off;
    const int16_t* H = Hboth + (size_t)s * (size_t)gd.dir[0].ncols * (size_t)L;
    const int16_t* sH = seeds + ((size_t)s * 2) * (size_t)n_nodes * (size_t)L;  // fill index = dir * 2 + strand = s
    const int16_t* sE = sH + (size_t)n_nodes * (size_t)L;
    auto qchar = [&](int j) -> uint32_t { return query_char(bases, L, 0, s, j); };
    auto Hcell = [&](uint32_t col, int j) -> int { return H[(size_t)col * (size_t)L + (size_t)j]; };
    auto seedHv = [&](uint32_t node, int j) -> int { return sH[(size_t)node * (size_t)L + (size_t)j]; };
    auto seedEv = [&](uint32_t node, int j) -> int { return sE[(size_t)node * (size_t)L + (size_t)j]; };

    Emitter em;
    em.slot = scratch;
    em.cap = pg_gen_ops_cap((uint32_t)L);
    em.n = 0;
    em.last_op = 0xFFu;
    em.last_node = 0;
    em.last_len = 0;
    em.overflow = false;
    em.node_has_ops = false;

    uint32_t n = (uint32_t)fs.max_node;
    int i = fs.ref_end, j = fs.read_end, sc = fs.score;
    bool inE = false, inF = false;
    int status = 0;
    uint32_t clipped = 0;
    if (L - 1 - j > 0)
    {
        em.emit(n, PG_OPC_S, (uint32_t)(L - 1 - j));
        clipped += (uint32_t)(L - 1 - j);
    }
    for (;;)
    {
        const PgNode nd = nodes[n];
        const uint32_t c0 = nd.col_start;
        // ---- inside the node (gssw.c:1214-1808) ------------------------------------------------------------------
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
                {  // open is tested before extend
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
                    em.emit(n, (rch == 'N' || qch == 'N') ? PG_OPC_N : (rch == qch ? PG_OPC_M : PG_OPC_X), 1);
                    sc -= sub;
                    --i;
                    --j;
                    continue;
                }
            }
            else if (sc == sub)
            {
                // the alignment starts on the first row / column: never an 'X' (gssw.c:1655-1690)
                if (rch == 'N' || qch == 'N')
                    em.emit(n, PG_OPC_N, 1);
                else if (rch == qch)
                    em.emit(n, PG_OPC_M, 1);
                sc -= sub;
                --i;
                --j;
                continue;
            }
            if (j > 0)
            {
                // score == F(i, j)  <=>  some k >= 1 with H(i, j - k) - go - (k - 1) ge == score
                bool isF = false;
                const int kmax = (j + 1 - sc - PG_GAP_OPEN + PG_GAP_EXT) / (1 + PG_GAP_EXT);
                const int lim = kmax < j ? kmax : j;
                for (int kk = 1; kk <= lim && !isF; ++kk)
                    isF = Hcell(c0 + i, j - kk) - PG_GAP_OPEN - (kk - 1) * PG_GAP_EXT == sc;
                if (isF)
                {
                    inF = true;
                    continue;
                }
            }
            if (i > 0)
            {
                inE = true;  // the only explanation left of H(i, j)
                continue;
            }
            // first column: E(0, j) is the seed, the maximum over the predecessors of their next-column E
            {
                int se = 0;
                for (uint32_t p = 0; p < nd.n_pred; ++p)
                    se = imax(se, seedEv(preds[nd.pred_off + p], j));
                if (sc == se)
                {
                    inE = true;
                    continue;
                }
            }
            break;  // a diagonal into a predecessor
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
        // ---- into a predecessor: the first one (ascending id) consistent with a diagonal / gap open / gap extend
        //      (gssw.c:2966-3161) ------------------------------------------------------------------------------------
        int best_prev = -1;
        for (uint32_t p = 0; p < nd.n_pred && best_prev < 0; ++p)
        {
            const uint32_t pid = preds[nd.pred_off + p];
            if (!inE)
            {
                if (j < 1)
                    continue;
                const uint32_t rch = (uint8_t)refc[c0 + i];
                const uint32_t qch = qchar(j);
                const int sub = sub_score(nt_code(rch), nt_code(qch));
                if (sc == seedHv(pid, j - 1) + sub)
                {
                    em.emit(n, (rch == 'N' || qch == 'N') ? PG_OPC_N : (rch == qch ? PG_OPC_M : PG_OPC_X), 1);
                    sc -= sub;
                    --j;
                    best_prev = (int)pid;
                }
            }
            else
            {
                if (sc == seedHv(pid, j) - PG_GAP_OPEN)
                {
                    em.emit(n, PG_OPC_D, 1);
                    sc += PG_GAP_OPEN;
                    inE = false;
                    best_prev = (int)pid;
                }
                else if (sc == seedEv(pid, j))
                {  // Enext_p = max(E_p - ge, H_p - go) and the open test failed  =>  score == E_p - ge
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
    *res_out = res;
    return em.n;
}
}  // namespace pggen
