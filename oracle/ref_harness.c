/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Drives the REFERENCE's own gssw.c (compiled in place from
 * /root/reference/external/gssw/gssw.c by oracle/Makefile into oracle/_ref/)
 * exactly the way the reference's GraphAligner does, so that the outputs of the
 * real reference arithmetic can be compared against (a) the plain-C restatement
 * in oracle/pg_oracle.c and (b) the HIP kernels.
 *
 * Only the glue is restated here (GraphAligner.cpp cannot be compiled in this
 * image: it needs Boost + spdlog); every line of DP / traceback arithmetic that
 * runs is the reference's.  Glue follows:
 *   initializeGraph        src/c++/lib/grm/GraphAligner.cpp:110-167
 *   setGraph/reverseGraph  GraphAligner.cpp:277-285, graph-tools GraphOperations.cpp:38-60
 *   alignsEndAtMultNodes   GraphAligner.cpp:170-212
 *   alignString            GraphAligner.cpp:214-227
 *   alignRead (4 fills, strand pick, mapq)   GraphAligner.cpp:308-404
 *   extractCigar           GraphAligner.cpp:88-108
 *   reverseComplement      graph-tools SequenceOperations.cpp:66-88
 */
#include <ctype.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gssw.h"

#define PGREF_AF_CIGAR 1u
#define PGREF_AF_BOTH_STRANDS 2u
#define PGREF_AF_REVERSE_GRAPH 4u

typedef struct
{
    uint32_t n_nodes;
    gssw_graph* g[2];     /* [0] forward graph, [1] reversed graph */
    gssw_node** nodes[2]; /* owned by g[] */
    int8_t* nt_table;
    int8_t* mat;
} pgref_graph;

typedef struct
{
    int32_t graph_pos;
    int32_t score;
    int32_t mapq;
    int32_t unique;
    int32_t returned_reverse;
    int32_t multi[4];  /* fwd-graph/fwd-strand, fwd-graph/rc-strand, rev-graph/fwd, rev-graph/rc */
    int32_t scores[4]; /* gssw_graph_mapping.score of each fill */
    int32_t cigar_len; /* strlen of the rendered graph CIGAR (may exceed the caller's buffer) */
} pgref_result;

typedef struct
{
    int32_t score;
    int32_t position;
    int32_t max_node;
    int32_t ref_end;
    int32_t read_end;
    int32_t multi;
    int32_t cigar_len;
} pgref_fill_result;

static gssw_graph* build_one(
    pgref_graph* pg, int dir, uint32_t n, const uint32_t* seq_off, const char* seq, const uint32_t* pred_off,
    const uint32_t* pred)
{
    /* dir 0: nodes as given.  dir 1: node i -> n-1-i, sequence reversed (not complemented), edges
     * reversed (GraphOperations.cpp:38-60); predecessors of a node are visited in ascending id
     * (std::set, Graph.hh:46) in both cases. */
    gssw_node** nodes = (gssw_node**)calloc(n, sizeof(gssw_node*));
    for (uint32_t id = 0; id < n; ++id)
    {
        uint32_t src = dir ? n - 1 - id : id;
        uint32_t len = seq_off[src + 1] - seq_off[src];
        char* s = (char*)malloc(len + 1);
        for (uint32_t k = 0; k < len; ++k)
        {
            char c = dir ? seq[seq_off[src] + len - 1 - k] : seq[seq_off[src] + k];
            s[k] = (char)toupper((unsigned char)c);
        }
        s[len] = 0;
        nodes[id] = gssw_node_create(NULL, id, s, pg->nt_table, pg->mat);
        free(s);
    }
    for (uint32_t to = 0; to < n; ++to)
    {
        if (!dir)
        {
            for (uint32_t k = pred_off[to]; k < pred_off[to + 1]; ++k)
            {
                gssw_nodes_add_edge(nodes[pred[k]], nodes[to]);
            }
        }
        else
        {
            /* predecessors of `to` in the reversed graph = {n-1-s : s successor of n-1-to in the
             * original}, ascending -> scan original nodes `s` descending. */
            uint32_t orig = n - 1 - to;
            for (int64_t s = (int64_t)n - 1; s >= 0; --s)
            {
                for (uint32_t k = pred_off[s]; k < pred_off[s + 1]; ++k)
                {
                    if (pred[k] == orig)
                    {
                        gssw_nodes_add_edge(nodes[n - 1 - (uint32_t)s], nodes[to]);
                    }
                }
            }
        }
    }
    gssw_graph* g = gssw_graph_create(n);
    for (uint32_t id = 0; id < n; ++id)
    {
        gssw_graph_add_node(g, nodes[id]);
    }
    pg->nodes[dir] = nodes;
    return g;
}

pgref_graph* pgref_graph_create(
    uint32_t n_nodes, const uint32_t* seq_off, const char* seq, const uint32_t* pred_off, const uint32_t* pred)
{
    pgref_graph* pg = (pgref_graph*)calloc(1, sizeof(pgref_graph));
    pg->n_nodes = n_nodes;
    pg->nt_table = gssw_create_nt_table();
    pg->mat = gssw_create_score_matrix(1, 4); /* GraphAligner.cpp:229-231 */
    pg->g[0] = build_one(pg, 0, n_nodes, seq_off, seq, pred_off, pred);
    pg->g[1] = build_one(pg, 1, n_nodes, seq_off, seq, pred_off, pred);
    return pg;
}

void pgref_graph_destroy(pgref_graph* pg)
{
    if (!pg)
        return;
    for (int d = 0; d < 2; ++d)
    {
        gssw_graph_destroy(pg->g[d]);
        free(pg->nodes[d]);
    }
    free(pg->nt_table);
    free(pg->mat);
    free(pg);
}

/* GraphAligner.cpp:170-212 (node_map is the identity: sequence expansion is off / trivial). */
static int aligns_end_at_mult_nodes(gssw_graph* graph, gssw_node** nodes, uint32_t n_nodes, int32_t read_len)
{
    uint16_t top_score = graph->max_node->alignment->score1;
    uint32_t hits = 0;
    for (uint32_t i = 0; i < n_nodes; ++i)
    {
        uint8_t* mH = (uint8_t*)nodes[i]->alignment->mH;
        int found = 0;
        for (int32_t row = 0; row < nodes[i]->len && !found; ++row)
        {
            for (int32_t col = 0; col < read_len; ++col)
            {
                uint8_t score = mH[read_len * row + col];
                if (score == top_score)
                {
                    found = 1;
                    break;
                }
            }
        }
        hits += (uint32_t)found;
        if (hits > 1)
            return 1;
    }
    return 0;
}

static int render_cigar(gssw_graph_mapping* gm, char* buf, int cap)
{
    /* GraphAligner.cpp:88-108 */
    int len = 0;
    gssw_graph_cigar* g = &gm->cigar;
    gssw_node_cigar* nc = g->elements;
    char tmp[64];
    for (uint32_t i = 0; i < g->length; ++i, ++nc)
    {
        int k = snprintf(tmp, sizeof tmp, "%u[", nc->node->id);
        if (buf && len + k < cap)
            memcpy(buf + len, tmp, (size_t)k);
        len += k;
        gssw_cigar* c = nc->cigar;
        gssw_cigar_element* e = c->elements;
        for (int32_t j = 0; j < c->length; ++j, ++e)
        {
            k = snprintf(tmp, sizeof tmp, "%u%c", e->length, e->type);
            if (buf && len + k < cap)
                memcpy(buf + len, tmp, (size_t)k);
            len += k;
        }
        if (buf && len + 1 < cap)
            buf[len] = ']';
        len += 1;
    }
    if (buf && cap > 0)
        buf[len < cap ? len : cap - 1] = 0;
    return len;
}

/* GraphAligner.cpp:214-227 */
static gssw_graph_mapping* align_string(pgref_graph* pg, int dir, const char* str_in, int len, int* multi)
{
    char* str = (char*)malloc((size_t)len + 1);
    for (int i = 0; i < len; ++i)
        str[i] = (char)toupper((unsigned char)str_in[i]);
    str[len] = 0;
    gssw_graph_fill(pg->g[dir], str, pg->nt_table, pg->mat, 6, 1, 15, 2);
    gssw_graph_mapping* gm = gssw_graph_trace_back(pg->g[dir], str, len, pg->nt_table, pg->mat, 6, 1);
    *multi = aligns_end_at_mult_nodes(pg->g[dir], pg->nodes[dir], pg->n_nodes, len);
    free(str);
    return gm;
}

static char complement_base(char b)
{
    switch (b)
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

/* One fill + traceback on graph direction `dir` (0 forward, 1 reversed): exposes the raw
 * gssw outputs (for kernel-level parity tests). `str` is used as given (upper-cased inside). */
int pgref_fill(
    pgref_graph* pg, int dir, const char* str, int len, pgref_fill_result* out, char* cigar_buf, int cigar_cap)
{
    int multi = 0;
    gssw_graph_mapping* gm = align_string(pg, dir, str, len, &multi);
    gssw_node* mn = pg->g[dir]->max_node;
    out->score = gm->score;
    out->position = gm->position;
    out->max_node = (int32_t)mn->id;
    out->ref_end = mn->alignment->ref_end1;
    out->read_end = mn->alignment->read_end1;
    out->multi = multi;
    out->cigar_len = render_cigar(gm, cigar_buf, cigar_cap);
    gssw_graph_mapping_destroy(gm);
    return 0;
}

/* Copies node `node`'s de-striped H/E/F byte matrices of the LAST fill on `dir` (row-major by
 * reference column, stride read_len) into caller buffers of size node_len*read_len. */
int pgref_dump_matrices(pgref_graph* pg, int dir, uint32_t node, int read_len, uint8_t* H, uint8_t* E, uint8_t* F)
{
    gssw_node* n = pg->nodes[dir][node];
    if (!n->alignment || !n->alignment->is_byte)
        return -1;
    size_t sz = (size_t)n->len * (size_t)read_len;
    if (H)
        memcpy(H, n->alignment->mH, sz);
    if (E)
        memcpy(E, n->alignment->mE, sz);
    if (F)
        memcpy(F, n->alignment->mF, sz);
    return 0;
}

/* GraphAligner::alignRead, GraphAligner.cpp:308-404.  `bases` is the read as stored in
 * common::Read (not upper-cased: reverseComplement sees the raw characters). */
int pgref_align_read(
    pgref_graph* pg, const char* bases, int len, unsigned flags, pgref_result* out, char* cigar_buf, int cigar_cap)
{
    char* rc = (char*)malloc((size_t)len + 1);
    char* rev = (char*)malloc((size_t)len + 1);
    char* rcrev = (char*)malloc((size_t)len + 1);
    for (int i = 0; i < len; ++i)
    {
        rc[i] = complement_base(bases[len - 1 - i]); /* reverseComplement(bases) */
        rev[i] = bases[len - 1 - i];                 /* std::reverse(bases) */
    }
    for (int i = 0; i < len; ++i)
        rcrev[i] = complement_base(rev[len - 1 - i]); /* reverseComplement(bases_rev) */
    rc[len] = rev[len] = rcrev[len] = 0;

    gssw_graph_mapping* gm[4] = { NULL, NULL, NULL, NULL };
    int multi[4] = { 0, 0, 0, 0 };
    gm[0] = align_string(pg, 0, bases, len, &multi[0]);
    if (flags & PGREF_AF_BOTH_STRANDS)
        gm[1] = align_string(pg, 0, rc, len, &multi[1]);
    if (flags & PGREF_AF_REVERSE_GRAPH)
    {
        gm[2] = align_string(pg, 1, rev, len, &multi[2]);
        if (flags & PGREF_AF_BOTH_STRANDS)
            gm[3] = align_string(pg, 1, rcrev, len, &multi[3]);
    }
    int fwd_unique = !multi[0] && !multi[2];
    int rev_unique = !multi[1] && !multi[3];
    int return_reverse = 0;
    if (!fwd_unique && rev_unique && gm[1])
        return_reverse = 1;
    else if (fwd_unique && !rev_unique)
        return_reverse = 0;
    else if (gm[1])
        return_reverse = gm[0]->score < gm[1]->score;

    gssw_graph_mapping* chosen = return_reverse ? gm[1] : gm[0];
    int unique = return_reverse ? rev_unique : fwd_unique;
    out->graph_pos = chosen->position;
    out->score = chosen->score;
    out->unique = unique;
    out->mapq = unique ? 60 : 0;
    out->returned_reverse = return_reverse;
    for (int k = 0; k < 4; ++k)
    {
        out->multi[k] = multi[k];
        out->scores[k] = gm[k] ? gm[k]->score : -1;
    }
    out->cigar_len = (flags & PGREF_AF_CIGAR) ? render_cigar(chosen, cigar_buf, cigar_cap) : 0;
    if (!(flags & PGREF_AF_CIGAR) && cigar_buf && cigar_cap > 0)
        cigar_buf[0] = 0;
    for (int k = 0; k < 4; ++k)
        if (gm[k])
            gssw_graph_mapping_destroy(gm[k]);
    free(rc);
    free(rev);
    free(rcrev);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Threaded batch driver = the reference's CPU parallelisation (Align.cpp:114-156): the read
 * vector is cut into `threads` contiguous chunks, each chunk gets its OWN aligner (own gssw
 * graphs, Align.cpp:107-110) and aligns its reads sequentially.  Used as bench.py's
 * cpu_baseline (kind "reference").
 * ---------------------------------------------------------------------------------------- */
typedef struct
{
    uint32_t n_nodes;
    const uint32_t* seq_off;
    const char* seq;
    const uint32_t* pred_off;
    const uint32_t* pred;
    const uint32_t* base_off;
    const char* bases;
    uint32_t begin, end;
    unsigned flags;
    pgref_result* results;
    char* cigars; /* n_reads * cigar_stride, may be NULL */
    int cigar_stride;
} pgref_chunk;

static void* chunk_main(void* p)
{
    pgref_chunk* c = (pgref_chunk*)p;
    pgref_graph* pg = pgref_graph_create(c->n_nodes, c->seq_off, c->seq, c->pred_off, c->pred);
    for (uint32_t r = c->begin; r < c->end; ++r)
    {
        int len = (int)(c->base_off[r + 1] - c->base_off[r]);
        char* cb = c->cigars ? c->cigars + (size_t)r * (size_t)c->cigar_stride : NULL;
        pgref_align_read(pg, c->bases + c->base_off[r], len, c->flags, &c->results[r], cb, c->cigars ? c->cigar_stride : 0);
    }
    pgref_graph_destroy(pg);
    return NULL;
}

int pgref_align_batch(
    uint32_t n_nodes, const uint32_t* seq_off, const char* seq, const uint32_t* pred_off, const uint32_t* pred,
    uint32_t n_reads, const uint32_t* base_off, const char* bases, unsigned flags, uint32_t threads,
    pgref_result* results, char* cigars, int cigar_stride)
{
    if (threads < 1)
        threads = 1;
    if (threads > n_reads && n_reads > 0)
        threads = n_reads;
    uint32_t step = n_reads ? (n_reads + threads - 1) / threads : 1;
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    pgref_chunk* ch = (pgref_chunk*)calloc(threads, sizeof(pgref_chunk));
    uint32_t used = 0;
    for (uint32_t t = 0; t < threads; ++t)
    {
        uint32_t b = t * step, e = b + step > n_reads ? n_reads : b + step;
        if (b >= e)
            break;
        ch[t] = (pgref_chunk){ n_nodes, seq_off, seq,   pred_off, pred,   base_off,    bases,
                               b,       e,       flags, results,  cigars, cigar_stride };
        pthread_create(&th[t], NULL, chunk_main, &ch[t]);
        ++used;
    }
    for (uint32_t t = 0; t < used; ++t)
        pthread_join(th[t], NULL);
    free(th);
    free(ch);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * KlibAligner stage: the reference's ksw.c (ksw_align + ksw_global, compiled in place) under the restated
 * wrapper of klib_glue.h.  KlibAlignment::update  src/c++/lib/common/Klib.cpp:144-164.
 * ---------------------------------------------------------------------------------------------- */
#include "ksw.h"
#include "klib_glue.h"

static void pgref_klib_engine(int qlen, uint8_t* query, int tlen, uint8_t* target, const int8_t* mat, int gapo, int gape, klib_pair* out)
{
    kswq_t* qp = 0;
    kswr_t r = ksw_align(qlen, query, tlen, target, 5, mat, gapo, gape, KSW_XSTART, &qp);
    free(qp);
    out->score = r.score;
    out->tb = r.tb;
    out->te = r.te;
    out->qb = r.qb;
    out->qe = r.qe;
    out->n_cigar = 0;
    out->cigar = 0;
    out->ub = (r.tb < 0 || r.qb < 0);
    if (out->ub)
        return;
    ksw_global(r.qe - r.qb + 1, query + r.qb, r.te - r.tb + 1, target + r.tb, 5, mat, gapo, gape, tlen, &out->n_cigar, &out->cigar);
}

/* one KlibAlignment (setRef / setQuery / getCigar); returns n_cigar (cigar: up to cap entries), out5 = score,r0,r1,a0,a1 */
int pgref_klib_pair(const char* ref, const char* query, int match, int mismatch, int gapo, int gape, int32_t* out5, uint32_t* cigar, int cap)
{
    const int tl = (int)strlen(ref), ql = (int)strlen(query);
    uint8_t* t = (uint8_t*)malloc((size_t)tl + 1);
    uint8_t* q = (uint8_t*)malloc((size_t)ql + 1);
    klib_translate(ref, t, tl);
    klib_translate(query, q, ql);
    int8_t mat[25];
    klib_matrix(mat, match, mismatch);
    klib_pair pr;
    memset(&pr, 0, sizeof pr);
    pgref_klib_engine(ql, q, tl, t, mat, gapo, gape, &pr);
    out5[0] = pr.score;
    out5[1] = pr.tb;
    out5[2] = pr.te;
    out5[3] = pr.qb;
    out5[4] = pr.qe;
    for (int i = 0; i < pr.n_cigar && i < cap; ++i)
        cigar[i] = pr.cigar[i];
    const int n = pr.ub ? -1 : pr.n_cigar;
    free(pr.cigar);
    free(t);
    free(q);
    return n;
}

int pgref_klib_align(
    int n_nodes, const uint32_t* node_off, const char* node_seq, int n_paths, const uint32_t* path_node_off, const uint32_t* path_nodes,
    uint32_t n_reads, const uint32_t* read_off, const char* read_bases, const uint8_t* bam_reverse, int match, int mismatch, int gapo,
    int gape, klib_result* results, char* cigars, int stride)
{
    return klib_align_batch(
        pgref_klib_engine, n_nodes, node_off, node_seq, n_paths, path_node_off, path_nodes, n_reads, read_off, read_bases, bam_reverse,
        match, mismatch, gapo, gape, results, cigars, stride);
}
