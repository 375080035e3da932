/* klib_glue.h -- TEST INFRASTRUCTURE ONLY (shared by oracle/ref_harness.c and oracle/pg_oracle.c).
 *
 * Restates the control flow AROUND the pairwise aligner of the reference's KlibAligner stage, parameterised by an
 * "engine" that performs one KlibAlignment::update():
 *
 *   common::KlibAlignment::{setRef,setQuery,update,getCigar}   src/c++/lib/common/Klib.cpp:72-164
 *   translate()                                               src/c++/lib/common/KlibImpl.hh:77-102
 *   common::AlignmentParameters                               src/c++/include/common/Alignment.hh:41-58
 *   grm::KlibAlignerImpl::{setGraph,align,alignRead}          src/c++/lib/grm/KlibAligner.cpp:186-205, 388-442
 *   grm::KlibAlignerImpl::{buildGraphCigar,updateAlignment,pickBest}   KlibAligner.cpp:207-308, 310-343, 349-386
 *   common::makeCigarBit / getCigarOp                         src/c++/lib/common/Alignment.cpp:72-114
 *   std::push_heap / std::pop_heap / std::min_element         libstdc++ (element ORDER decides the reported candidate)
 *
 * With the engine bound to the reference's own ksw_align/ksw_global (ref_harness.c) this is "reference ksw + restated
 * wrapper" (the KlibAligner translation unit itself needs boost/spdlog headers the image lacks); with the engine bound
 * to the scalar restatement of ksw (pg_oracle.c) it is the travelling port.  Both are pinned on
 * src/c++/test/test_klibaligner.cpp:149-193 and src/c++/test/test_align.cpp:38-263 (tests/test_klib_oracle.py).
 *
 * One deliberate deviation: when ksw_align's reverse pass does not reproduce the forward score it leaves tb = qb = -1
 * (ksw.c:351-352) and the reference then reads one byte before its buffers (Klib.cpp:155-159) -- undefined behaviour.
 * Here such a (path, strand) produces no candidate and the read is flagged `ub` so tests can tell.
 */
#ifndef PG_KLIB_GLUE_H
#define PG_KLIB_GLUE_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
    int score, tb, te, qb, qe;
    int n_cigar;
    uint32_t* cigar; /* BAM encoding len<<4|op (0 M, 1 I, 2 D); malloc'd by the engine, freed by the caller */
    int ub;
} klib_pair;

/* one KlibAlignment::update(): local SW with start recovery + banded global over the local window (band = tlen) */
typedef void (*klib_engine)(int qlen, uint8_t* query, int tlen, uint8_t* target, const int8_t* mat, int gapo, int gape, klib_pair* out);

typedef struct
{
    int32_t status; /* 0 UNMAPPED, 1 MAPPED, 2 BAD_ALIGN */
    int32_t graph_pos, score, mapq, unique, is_graph_reverse, used_reverse, ub, cigar_len;
} klib_result;

static inline void klib_translate(const char* s, uint8_t* out, int n)
{ /* KlibImpl.hh:77-102: A/a/U/u -> 0, C/c -> 1, G/g -> 2, T/t -> 3, everything else 4 (index masked with 0x7f) */
    for (int i = 0; i < n; ++i)
    {
        switch (((int)s[i]) & 0x7f)
        {
        case 'A': case 'a': case 'U': case 'u': out[i] = 0; break;
        case 'C': case 'c': out[i] = 1; break;
        case 'G': case 'g': out[i] = 2; break;
        case 'T': case 't': out[i] = 3; break;
        default: out[i] = 4;
        }
    }
}

static inline void klib_matrix(int8_t* mat, int match, int mismatch)
{ /* Alignment.hh:46-55 */
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            mat[a * 5 + b] = (a == 4 || b == 4) ? 0 : (a == b ? (int8_t)match : (int8_t)mismatch);
}

static inline char klib_comp(char c)
{ /* graph-tools SequenceOperations.cpp:66-81 */
    switch (c)
    {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}

typedef struct
{
    int path, reverse, score;
    size_t pos;
    int n_cigar;
    uint32_t* cigar; /* path CIGAR incl. soft clips, op 4 = S */
} klib_cand;

/* Candidate::betterScore as the heap's "less" */
static inline int klib_less(const klib_cand* a, const klib_cand* b) { return a->score > b->score; }

static void klib_push_heap(klib_cand* a, int n)
{
    klib_cand value = a[n - 1];
    int hole = n - 1, parent = (hole - 1) / 2;
    while (hole > 0 && klib_less(&a[parent], &value))
    {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[hole] = value;
}

static void klib_pop_heap(klib_cand* a, int n)
{ /* libstdc++ __pop_heap + __adjust_heap: afterwards a[n-1] holds the old top */
    klib_cand value = a[n - 1];
    a[n - 1] = a[0];
    const int len = n - 1;
    int hole = 0, second = 0;
    while (second < (len - 1) / 2)
    {
        second = 2 * (second + 1);
        if (klib_less(&a[second], &a[second - 1]))
            --second;
        a[hole] = a[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2)
    {
        second = 2 * (second + 1);
        a[hole] = a[second - 1];
        hole = second - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > 0 && klib_less(&a[parent], &value))
    {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    if (len > 0)
        a[hole] = value;
}

typedef struct
{
    int n_nodes;
    const uint32_t* node_off; /* into node_seq, n_nodes + 1 */
    const char* node_seq;
    int n_paths;
    const uint32_t* path_node_off; /* n_paths + 1 */
    const uint32_t* path_nodes;
    /* derived */
    char** pseq;
    uint8_t** pcode;
    int* plen;
} klib_graph;

static void klib_graph_prepare(klib_graph* g)
{
    g->pseq = (char**)calloc((size_t)g->n_paths + 1, sizeof(char*));
    g->pcode = (uint8_t**)calloc((size_t)g->n_paths + 1, sizeof(uint8_t*));
    g->plen = (int*)calloc((size_t)g->n_paths + 1, sizeof(int));
    for (int p = 0; p < g->n_paths; ++p)
    {
        int len = 0;
        for (uint32_t q = g->path_node_off[p]; q < g->path_node_off[p + 1]; ++q)
            len += (int)(g->node_off[g->path_nodes[q] + 1] - g->node_off[g->path_nodes[q]]);
        g->pseq[p] = (char*)malloc((size_t)len + 1);
        g->pcode[p] = (uint8_t*)malloc((size_t)len + 1);
        int at = 0;
        for (uint32_t q = g->path_node_off[p]; q < g->path_node_off[p + 1]; ++q)
        {
            const uint32_t nd = g->path_nodes[q];
            const int l = (int)(g->node_off[nd + 1] - g->node_off[nd]);
            memcpy(g->pseq[p] + at, g->node_seq + g->node_off[nd], (size_t)l);
            at += l;
        }
        g->pseq[p][len] = 0;
        g->plen[p] = len;
        klib_translate(g->pseq[p], g->pcode[p], len);
    }
}

static void klib_graph_release(klib_graph* g)
{
    for (int p = 0; p < g->n_paths; ++p)
    {
        free(g->pseq[p]);
        free(g->pcode[p]);
    }
    free(g->pseq);
    free(g->pcode);
    free(g->plen);
}

static int klib_node_len(const klib_graph* g, uint32_t nd) { return (int)(g->node_off[nd + 1] - g->node_off[nd]); }

/* buildGraphCigar (KlibAligner.cpp:207-308): returns graph_pos, writes the CIGAR string and the match count */
static int klib_graph_cigar(const klib_graph* g, const klib_cand* c, const char* seq, char* out, size_t cap, int* matches)
{
    const int p = c->path;
    /* findStartNode: the path node whose start is the last one <= position (map lower_bound logic, :112-127) */
    uint32_t qi = g->path_node_off[p];
    size_t node_first = 0;
    {
        size_t start = 0;
        for (uint32_t q = g->path_node_off[p]; q < g->path_node_off[p + 1]; ++q)
        {
            if (start <= c->pos)
            {
                qi = q;
                node_first = start;
            }
            start += (size_t)klib_node_len(g, g->path_nodes[q]);
        }
    }
    size_t node_pos = c->pos - node_first;
    const int ret = (int)node_pos;
    size_t n = 0;
    *matches = 0;
#define KL_APPEND(...)                                                        \
    do                                                                        \
    {                                                                         \
        if (n < cap)                                                          \
            n += (size_t)snprintf(out + n, cap - n, __VA_ARGS__);             \
    } while (0)
    KL_APPEND("%u[", g->path_nodes[qi]);
    const char* it = seq;
    for (int e = 0; e < c->n_cigar; ++e)
    {
        const uint32_t code = c->cigar[e] & 0xf;
        size_t len = c->cigar[e] >> 4;
        if (code == 0)
        {
            while (len)
            {
                const size_t room = (size_t)klib_node_len(g, g->path_nodes[qi]) - node_pos;
                const size_t piece = len < room ? len : room;
                /* makeCigarBit */
                const char* ref = g->pseq[p] + node_pos + node_first;
                char last = 0;
                size_t last_len = 0;
                for (size_t k = 0; k < piece; ++k, ++it, ++last_len)
                {
                    const char s = ref[k], r = *it;
                    const char op = s == r ? 'M' : (s == 'N' ? 'N' : (r == 'N' ? 'N' : 'X'));
                    if (op != last)
                    {
                        if (last_len)
                        {
                            KL_APPEND("%zu%c", last_len, last);
                            if (last == 'M')
                                *matches += (int)last_len;
                        }
                        last = op;
                        last_len = 0;
                    }
                }
                if (last_len)
                {
                    KL_APPEND("%zu%c", last_len, last);
                    if (last == 'M')
                        *matches += (int)last_len;
                }
                len -= piece;
                if (len)
                {
                    node_first += (size_t)klib_node_len(g, g->path_nodes[qi]);
                    ++qi;
                    node_pos = 0;
                    KL_APPEND("]%u[", g->path_nodes[qi]);
                }
                else
                    node_pos += piece;
            }
        }
        else if (code == 2)
        {
            while (len)
            {
                const size_t room = (size_t)klib_node_len(g, g->path_nodes[qi]) - node_pos;
                const size_t piece = len < room ? len : room;
                if (piece)
                    KL_APPEND("%zuD", piece);
                len -= piece;
                if (len)
                {
                    node_first += (size_t)klib_node_len(g, g->path_nodes[qi]);
                    ++qi;
                    node_pos = 0;
                    KL_APPEND("]%u[", g->path_nodes[qi]);
                }
                else
                    node_pos += piece;
            }
        }
        else if (code == 1)
        {
            KL_APPEND("%zuI", len);
            it += len;
        }
        else
        {
            KL_APPEND("%zuS", len);
            it += len;
        }
    }
    KL_APPEND("]");
#undef KL_APPEND
    if (n >= cap)
        n = cap - 1;
    out[n] = 0;
    return ret;
}

static int klib_min_element(const klib_cand* a, int from, int n)
{ /* std::min_element(first, last, betterScore): first element no other is "better" than */
    if (from >= n)
        return n;
    int best = from;
    for (int i = from + 1; i < n; ++i)
        if (klib_less(&a[i], &a[best]))
            best = i;
    return best;
}

/* KlibAlignerImpl::alignRead for one read. cigar_out: cap bytes. */
static void klib_align_read(
    klib_engine engine, const klib_graph* g, const char* bases, int L, int bam_reverse, int match, int mismatch, int gapo, int gape,
    klib_result* res, char* cigar_out, size_t cap)
{
    memset(res, 0, sizeof *res);
    cigar_out[0] = 0;
    int8_t mat[25];
    klib_matrix(mat, match, mismatch);
    char* rv = (char*)malloc((size_t)L + 1);
    char* fw = (char*)malloc((size_t)L + 1);
    memcpy(fw, bases, (size_t)L);
    fw[L] = 0;
    for (int i = 0; i < L; ++i)
        rv[i] = klib_comp(bases[L - 1 - i]);
    rv[L] = 0;
    uint8_t* code = (uint8_t*)malloc((size_t)L + 1);
    const int cap_c = g->n_paths + 2;
    klib_cand* cands = (klib_cand*)calloc((size_t)cap_c, sizeof(klib_cand));
    int nc = 0;
    for (int p = 0; p < g->n_paths; ++p)
    {
        for (int reverse = 0; reverse < 2; ++reverse)
        {
            const char* seq = reverse ? rv : fw;
            klib_translate(seq, code, L);
            klib_pair pr;
            memset(&pr, 0, sizeof pr);
            uint8_t* tcopy = (uint8_t*)malloc((size_t)g->plen[p] + 1);
            memcpy(tcopy, g->pcode[p], (size_t)g->plen[p]);
            engine(L, code, g->plen[p], tcopy, mat, gapo, gape, &pr);
            free(tcopy);
            if (pr.ub)
            {
                res->ub = 1;
                free(pr.cigar);
                continue;
            }
            if (pr.te < pr.tb)
            { /* fully soft clipped alignment. Ignore (:404-408) */
                free(pr.cigar);
                continue;
            }
            klib_cand c;
            c.path = p;
            c.reverse = reverse;
            c.score = pr.score;
            c.pos = (size_t)pr.tb;
            c.cigar = (uint32_t*)malloc(((size_t)pr.n_cigar + 2) * sizeof(uint32_t));
            c.n_cigar = 0;
            if (pr.qb)
                c.cigar[c.n_cigar++] = ((uint32_t)pr.qb << 4) | 4u;
            for (int k = 0; k < pr.n_cigar; ++k)
                c.cigar[c.n_cigar++] = pr.cigar[k];
            const uint32_t right = (uint32_t)(L - pr.qe - 1);
            if (right)
                c.cigar[c.n_cigar++] = (right << 4) | 4u;
            free(pr.cigar);
            cands[nc++] = c;
            klib_push_heap(cands, nc);
            if (nc == cap_c)
            {
                klib_pop_heap(cands, nc);
                free(cands[nc - 1].cigar);
                --nc;
            }
        }
    }
    if (nc)
    { /* pickBest (:349-386) */
        const int bi = klib_min_element(cands, 0, nc);
        const klib_cand* best = &cands[bi];
        int matches = 0;
        res->graph_pos = klib_graph_cigar(g, best, best->reverse ? rv : fw, cigar_out, cap, &matches);
        res->score = matches;
        res->used_reverse = best->reverse;
        res->is_graph_reverse = best->reverse ? !bam_reverse : bam_reverse;
        res->mapq = 60;
        res->unique = 1;
        res->status = 1;
        char* other = (char*)malloc(cap);
        for (int si = klib_min_element(cands, bi + 1, nc); si != nc; si = klib_min_element(cands, si + 1, nc))
        {
            const klib_cand* sb = &cands[si];
            if (sb->score != best->score)
                break;
            int m2 = 0;
            const int pos2 = klib_graph_cigar(g, sb, sb->reverse ? rv : fw, other, cap, &m2);
            if (strcmp(other, cigar_out) != 0 || pos2 != res->graph_pos)
            {
                res->mapq = 0;
                res->unique = 0;
                res->status = 2;
                break;
            }
        }
        free(other);
    }
    res->cigar_len = (int32_t)strlen(cigar_out);
    for (int i = 0; i < nc; ++i)
        free(cands[i].cigar);
    free(cands);
    free(code);
    free(rv);
    free(fw);
}

/* Batch entry shared by both libraries. */
static int klib_align_batch(
    klib_engine engine, int n_nodes, const uint32_t* node_off, const char* node_seq, int n_paths, const uint32_t* path_node_off,
    const uint32_t* path_nodes, uint32_t n_reads, const uint32_t* read_off, const char* read_bases, const uint8_t* bam_reverse,
    int match, int mismatch, int gapo, int gape, klib_result* results, char* cigars, int stride)
{
    klib_graph g;
    memset(&g, 0, sizeof g);
    g.n_nodes = n_nodes;
    g.node_off = node_off;
    g.node_seq = node_seq;
    g.n_paths = n_paths;
    g.path_node_off = path_node_off;
    g.path_nodes = path_nodes;
    klib_graph_prepare(&g);
    for (uint32_t r = 0; r < n_reads; ++r)
        klib_align_read(
            engine, &g, read_bases + read_off[r], (int)(read_off[r + 1] - read_off[r]), bam_reverse ? bam_reverse[r] : 0, match,
            mismatch, gapo, gape, &results[r], cigars + (size_t)r * (size_t)stride, (size_t)stride);
    klib_graph_release(&g);
    return 0;
}
#endif
