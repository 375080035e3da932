/*
 * oracle/pg_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or shipped
 * with the product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load the library built from this file).
 *
 * Plain-C, scalar CPU restatement of the reference's read -> variant-graph realignment path:
 *
 *   score matrix / nt table      external/gssw/gssw.c:4188-4220
 *   per-node seeded affine fill  external/gssw/gssw.c:153-473 (gssw_sw_sse2_byte), 3897-3931 (seed max),
 *                                3963-4028 (graph fill, max_node rule)
 *   within-node traceback        external/gssw/gssw.c:1112-1818 (final_traceback = 1, no deflections)
 *   graph-level traceback        external/gssw/gssw.c:2621-3537
 *   CIGAR run-length encoding    external/gssw/gssw.c:3679-3746
 *   GraphAligner glue            src/c++/lib/grm/GraphAligner.cpp:88-108, 110-167, 170-227, 308-404
 *   reverseComplement            graph-tools src/graphutils/SequenceOperations.cpp:66-88
 *
 * The fill is written in the textbook (non-striped) form of SURVEY.md section 8(a'): gssw's striped
 * kernel produces the same H in every cell; its stored E/F matrices are <= the textbook values in a
 * few % of cells but never where a traceback decision reads them.  That claim is PINNED, not
 * assumed: tests/test_oracle_vs_ref.py compares this file against the reference's own gssw.c
 * (oracle/_ref/libpg_ref.so) on the reference's golden vectors and on randomized graphs/reads, and
 * tests/golden/ holds vectors generated from the real gssw.c.
 *
 * Scores are held in int32, i.e. no 8-bit saturation: identical to gssw for reads <= 250 bp
 * (250 + bias 4 < 255, gssw.c:380); longer reads take gssw's 16-bit restart path ("word mode") once a score
 * reaches 251, which is the same arithmetic -- except that GraphAligner's alignsEndAtMultNodes then reads the 16-bit
 * matrix through a byte pointer, which aligns_end_at_mult_nodes() below reproduces.  Pinned against the real gssw.c
 * by tests/test_oracle.py::test_port_vs_reference_gssw_word_mode.
 */
#include <ctype.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PGO_AF_CIGAR 1u
#define PGO_AF_BOTH_STRANDS 2u
#define PGO_AF_REVERSE_GRAPH 4u

#define GAP_OPEN 6
#define GAP_EXT 1

typedef struct
{
    int32_t len;
    char* seq;    /* upper-cased characters */
    int8_t* code; /* gssw nt codes */
    int32_t n_pred;
    int32_t* pred; /* ascending ids */
    /* per-fill state */
    int32_t* H; /* [len][L] */
    int32_t* E;
    int32_t* F;
    int32_t* Enext; /* [L] */
    int32_t score1, ref_end1, read_end1;
} pgo_node;

typedef struct
{
    uint32_t n_nodes;
    pgo_node* nodes[2]; /* [0] forward graph, [1] reversed graph */
    int32_t cap_L;
} pgo_graph;

typedef struct
{
    int32_t graph_pos;
    int32_t score;
    int32_t mapq;
    int32_t unique;
    int32_t returned_reverse;
    int32_t multi[4];
    int32_t scores[4];
    int32_t cigar_len;
} pgo_result;

typedef struct
{
    int32_t score;
    int32_t position;
    int32_t max_node;
    int32_t ref_end;
    int32_t read_end;
    int32_t multi;
    int32_t cigar_len;
} pgo_fill_result;

/* gssw.c:4206-4220 */
static int8_t nt_code(char c)
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

/* gssw.c:4188-4204 with match 1, mismatch 4 (GraphAligner.cpp:229-231) */
static inline int32_t sub_score(int8_t a, int8_t b)
{
    if (a == 4 || b == 4)
        return 0;
    return a == b ? 1 : -4;
}

static inline int32_t sat_sub(int32_t a, int32_t b) { return a > b ? a - b : 0; }
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

static void build_dir(
    pgo_graph* g, int dir, uint32_t n, const uint32_t* seq_off, const char* seq, const uint32_t* pred_off,
    const uint32_t* pred)
{
    pgo_node* nodes = (pgo_node*)calloc(n, sizeof(pgo_node));
    for (uint32_t id = 0; id < n; ++id)
    {
        uint32_t src = dir ? n - 1 - id : id;
        int32_t len = (int32_t)(seq_off[src + 1] - seq_off[src]);
        nodes[id].len = len;
        nodes[id].seq = (char*)malloc((size_t)len + 1);
        nodes[id].code = (int8_t*)malloc((size_t)len + 1);
        for (int32_t k = 0; k < len; ++k)
        {
            char c = dir ? seq[seq_off[src] + (uint32_t)(len - 1 - k)] : seq[seq_off[src] + (uint32_t)k];
            c = (char)toupper((unsigned char)c);
            nodes[id].seq[k] = c;
            nodes[id].code[k] = nt_code(c);
        }
        nodes[id].seq[len] = 0;
        /* predecessors, ascending id (Graph adjacency is std::set) */
        int32_t cnt = 0;
        int32_t* pl = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
        if (!dir)
        {
            for (uint32_t k = pred_off[id]; k < pred_off[id + 1]; ++k)
                pl[cnt++] = (int32_t)pred[k];
        }
        else
        {
            uint32_t orig = n - 1 - id;
            for (int64_t s = (int64_t)n - 1; s >= 0; --s)
                for (uint32_t k = pred_off[s]; k < pred_off[s + 1]; ++k)
                    if (pred[k] == orig)
                        pl[cnt++] = (int32_t)(n - 1 - (uint32_t)s);
        }
        nodes[id].n_pred = cnt;
        nodes[id].pred = pl;
    }
    g->nodes[dir] = nodes;
}

pgo_graph* pgo_graph_create(
    uint32_t n_nodes, const uint32_t* seq_off, const char* seq, const uint32_t* pred_off, const uint32_t* pred)
{
    pgo_graph* g = (pgo_graph*)calloc(1, sizeof(pgo_graph));
    g->n_nodes = n_nodes;
    build_dir(g, 0, n_nodes, seq_off, seq, pred_off, pred);
    build_dir(g, 1, n_nodes, seq_off, seq, pred_off, pred);
    return g;
}

void pgo_graph_destroy(pgo_graph* g)
{
    if (!g)
        return;
    for (int d = 0; d < 2; ++d)
    {
        for (uint32_t i = 0; i < g->n_nodes; ++i)
        {
            pgo_node* n = &g->nodes[d][i];
            free(n->seq);
            free(n->code);
            free(n->pred);
            free(n->H);
            free(n->E);
            free(n->F);
            free(n->Enext);
        }
        free(g->nodes[d]);
    }
    free(g);
}

static void ensure_capacity(pgo_graph* g, int32_t L)
{
    if (L <= g->cap_L)
        return;
    for (int d = 0; d < 2; ++d)
        for (uint32_t i = 0; i < g->n_nodes; ++i)
        {
            pgo_node* n = &g->nodes[d][i];
            size_t sz = sizeof(int32_t) * (size_t)(n->len ? n->len : 1) * (size_t)L;
            n->H = (int32_t*)realloc(n->H, sz);
            n->E = (int32_t*)realloc(n->E, sz);
            n->F = (int32_t*)realloc(n->F, sz);
            n->Enext = (int32_t*)realloc(n->Enext, sizeof(int32_t) * (size_t)L);
        }
    g->cap_L = L;
}

/* ---- fill: SURVEY 8(a'), gssw.c:153-473 + 3897-3931 + 3963-4028 -------------------------------- */
static int32_t fill_graph(pgo_graph* g, int dir, const int8_t* q, int32_t L)
{
    pgo_node* nodes = g->nodes[dir];
    int32_t* seedH = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L ? L : 1));
    int32_t* seedE = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L ? L : 1));
    int32_t max_node = -1, max_score = 0;
    for (uint32_t id = 0; id < g->n_nodes; ++id)
    {
        pgo_node* n = &nodes[id];
        for (int32_t j = 0; j < L; ++j)
        {
            int32_t sh = 0, se = 0;
            for (int32_t k = 0; k < n->n_pred; ++k)
            {
                pgo_node* p = &nodes[n->pred[k]];
                if (p->len > 0)
                    sh = imax(sh, p->H[(size_t)(p->len - 1) * (size_t)L + (size_t)j]);
                se = imax(se, p->Enext[j]);
            }
            seedH[j] = sh;
            seedE[j] = se;
        }
        int32_t best = 0, end_ref = -1, end_read = L - 1;
        for (int32_t i = 0; i < n->len; ++i)
        {
            int32_t* Hc = n->H + (size_t)i * (size_t)L;
            int32_t* Ec = n->E + (size_t)i * (size_t)L;
            int32_t* Fc = n->F + (size_t)i * (size_t)L;
            const int32_t* Hp = i > 0 ? Hc - L : seedH;
            int32_t colmax = 0;
            for (int32_t j = 0; j < L; ++j)
            {
                int32_t e = i > 0 ? imax(sat_sub(Ec[j - L], GAP_EXT), sat_sub(Hc[j - L], GAP_OPEN)) : seedE[j];
                int32_t f = j > 0 ? imax(sat_sub(Fc[j - 1], GAP_EXT), sat_sub(Hc[j - 1], GAP_OPEN)) : 0;
                int32_t diag = j > 0 ? Hp[j - 1] : 0;
                int32_t h = imax(diag + sub_score(n->code[i], q[j]), 0);
                h = imax(h, imax(e, f));
                Hc[j] = h;
                Ec[j] = e;
                Fc[j] = f;
                colmax = imax(colmax, h);
            }
            if (colmax > best)
            { /* first column reaching the node maximum (gssw.c:369-386) */
                best = colmax;
                end_ref = i;
            }
        }
        if (n->len > 0)
        {
            const int32_t* Hl = n->H + (size_t)(n->len - 1) * (size_t)L;
            const int32_t* El = n->E + (size_t)(n->len - 1) * (size_t)L;
            for (int32_t j = 0; j < L; ++j)
                n->Enext[j] = imax(sat_sub(El[j], GAP_EXT), sat_sub(Hl[j], GAP_OPEN));
        }
        else
        {
            /* zero-length node: gssw's loop does not run; the seed passes through unchanged
             * (gssw.c:214-218, 442-443) */
            for (int32_t j = 0; j < L; ++j)
                n->Enext[j] = seedE[j];
        }
        if (end_ref >= 0)
        { /* smallest read index holding the maximum in that column (gssw.c:446-454) */
            const int32_t* Hc = n->H + (size_t)end_ref * (size_t)L;
            for (int32_t j = 0; j < L; ++j)
                if (Hc[j] == best)
                {
                    end_read = j;
                    break;
                }
        }
        if (best == 0 && L > 0)
            end_read = 0; /* all-zero node: every entry of gssw's zero-initialised pvHmax "equals the
                             maximum", so its scan settles on read index 0 (gssw.c:446-454) */
        n->score1 = best;
        n->ref_end1 = end_ref;
        n->read_end1 = end_read;
        /* gssw.c:4015-4018; the stale-max_node leak for all-zero fills is not modelled (it cannot
         * change any output: see DESIGN.md "degenerate reads") */
        if (max_node < 0 || best > max_score)
        {
            max_node = (int32_t)id;
            max_score = best;
        }
    }
    free(seedH);
    free(seedE);
    return max_node;
}

/* ---- CIGAR containers -------------------------------------------------------------------------- */
typedef struct
{
    char type;
    uint32_t length;
} cig_el;
typedef struct
{
    int32_t n, cap;
    cig_el* el;
} cigar;
typedef struct
{
    int32_t node;
    cigar c;
} node_cigar;

static void cig_push_back(cigar* c, char type, uint32_t length)
{ /* gssw.c:3679-3695 */
    if (c->n > 0 && c->el[c->n - 1].type == type)
    {
        c->el[c->n - 1].length += length;
        return;
    }
    if (c->n == c->cap)
    {
        c->cap = c->cap ? c->cap * 2 : 8;
        c->el = (cig_el*)realloc(c->el, sizeof(cig_el) * (size_t)c->cap);
    }
    c->el[c->n].type = type;
    c->el[c->n].length = length;
    c->n++;
}
static void cig_reverse(cigar* c)
{
    for (int32_t s = 0, e = c->n - 1; s < e; ++s, --e)
    {
        cig_el t = c->el[s];
        c->el[s] = c->el[e];
        c->el[e] = t;
    }
}
static void cig_push_front(cigar* c, char type, uint32_t length)
{ /* gssw.c:3697-3701 */
    cig_reverse(c);
    cig_push_back(c, type, length);
    cig_reverse(c);
}

static char match_op(char r, char q)
{
    if (r == 'N' || q == 'N')
        return 'N';
    return r == q ? 'M' : 'X';
}

/* within-node traceback, gssw.c:1112-1818 with final_traceback = 1 and no deflections.
 * returns 0 ok, -1 inconsistent ("stuck": the reference would assert / spin). */
static int node_trace_back(
    const pgo_node* n, int32_t L, const char* read, const int8_t* q, int32_t* score, int32_t* refEnd,
    int32_t* readEnd, int32_t* gRefFlag, int32_t* gReadFlag, cigar* result)
{
    int32_t i = *refEnd, j = *readEnd, gRead = *gReadFlag, gRef = *gRefFlag;
    const int32_t* H = n->H;
    const int32_t* E = n->E;
    const int32_t* F = n->F;
#define AT(M, I, J) (M)[(size_t)(I) * (size_t)L + (size_t)(J)]
    int32_t sc = gRead ? AT(E, i, j) : (gRef ? AT(F, i, j) : AT(H, i, j));
    int rc = 0;
    while (sc > 0 && i >= 0 && j >= 0)
    {
        if (gRead)
        {
            if (i > 0)
            {
                if (sc == AT(H, i - 1, j) - GAP_OPEN)
                {
                    cig_push_back(result, 'D', 1);
                    sc += GAP_OPEN;
                    --i;
                    gRead = 0;
                    continue;
                }
                if (sc == AT(E, i - 1, j) - GAP_EXT)
                {
                    cig_push_back(result, 'D', 1);
                    sc += GAP_EXT;
                    --i;
                    continue;
                }
                rc = -1;
                break;
            }
            break; /* i == 0: read gap leaves through the left edge */
        }
        else if (gRef)
        {
            if (j > 0)
            {
                if (sc == AT(H, i, j - 1) - GAP_OPEN)
                {
                    cig_push_back(result, 'I', 1);
                    sc += GAP_OPEN;
                    --j;
                    gRef = 0;
                    continue;
                }
                if (sc == AT(F, i, j - 1) - GAP_EXT)
                {
                    cig_push_back(result, 'I', 1);
                    sc += GAP_EXT;
                    --j;
                    continue;
                }
            }
            rc = -1;
            break;
        }
        else
        {
            int32_t s = sub_score(n->code[i], q[j]);
            if (i > 0 && j > 0)
            {
                if (sc == AT(H, i - 1, j - 1) + s)
                {
                    cig_push_back(result, match_op(n->seq[i], read[j]), 1);
                    sc -= s;
                    --i;
                    --j;
                    continue;
                }
            }
            else if (sc == s)
            { /* alignment start on the first row / first column (gssw.c:1655-1690): an 'X' is never
                 emitted here */
                if (n->seq[i] == 'N' || read[j] == 'N')
                    cig_push_back(result, 'N', 1);
                else if (n->seq[i] == read[j])
                    cig_push_back(result, 'M', 1);
                sc -= s;
                --i;
                --j;
                continue;
            }
            if (j > 0 && sc == AT(F, i, j))
            {
                gRef = 1;
                continue;
            }
            if (sc == AT(E, i, j))
            {
                gRead = 1;
                continue;
            }
            if (i == 0)
                break; /* try a diagonal into a predecessor node */
            rc = -1;
            break;
        }
    }
#undef AT
    *score = sc;
    *refEnd = i;
    *readEnd = j;
    *gRefFlag = gRef;
    *gReadFlag = gRead;
    cig_reverse(result);
    return rc;
}

typedef struct
{
    int32_t score;
    int32_t position;
    int32_t n_nodes;
    node_cigar* nc;
    int status; /* 0 ok, -1 traceback inconsistent */
} mapping;

static void mapping_free(mapping* m)
{
    for (int32_t i = 0; i < m->n_nodes; ++i)
        free(m->nc[i].c.el);
    free(m->nc);
    m->nc = NULL;
    m->n_nodes = 0;
}

/* gssw.c:2621-3537, num_tracebacks = 1, no pinned node */
static mapping graph_trace_back(pgo_graph* g, int dir, int32_t max_node, const char* read, const int8_t* q, int32_t L)
{
    pgo_node* nodes = g->nodes[dir];
    mapping m;
    memset(&m, 0, sizeof m);
    int32_t cap = 16;
    m.nc = (node_cigar*)calloc((size_t)cap, sizeof(node_cigar));
    int32_t n = max_node;
    int32_t refEnd = nodes[n].ref_end1, readEnd = nodes[n].read_end1;
    m.score = nodes[n].score1;
    int32_t score;
    if (readEnd < 0 || refEnd < 0)
        score = 0;
    else
        score = nodes[n].H[(size_t)refEnd * (size_t)L + (size_t)readEnd];
    int32_t gapInRef = 0, gapInRead = 0;
    int32_t end_soft_clip = L - readEnd - 1;
    while (score > 0)
    {
        if (m.n_nodes == cap)
        {
            cap *= 2;
            m.nc = (node_cigar*)realloc(m.nc, sizeof(node_cigar) * (size_t)cap);
            memset(m.nc + m.n_nodes, 0, sizeof(node_cigar) * (size_t)(cap - m.n_nodes));
        }
        node_cigar* nc = &m.nc[m.n_nodes];
        nc->node = n;
        if (node_trace_back(&nodes[n], L, read, q, &score, &refEnd, &readEnd, &gapInRef, &gapInRead, &nc->c) != 0)
            m.status = -1;
        if (end_soft_clip)
        {
            cig_push_back(&nc->c, 'S', (uint32_t)end_soft_clip);
            end_soft_clip = 0;
        }
        ++m.n_nodes;
        if (m.status != 0)
            break;
        if (score != 0 && refEnd > 0)
        {
            m.status = -1;
            break;
        }
        if (score == 0)
        {
            if (readEnd > -1)
                cig_push_front(&nc->c, 'S', (uint32_t)(readEnd + 1));
            break;
        }
        /* cross into a predecessor: first one (ascending id) consistent with diagonal / gap open /
         * gap extend (gssw.c:2966-3161) */
        int32_t best_prev = -1;
        const pgo_node* cur = &nodes[n];
        for (int32_t k = 0; k < cur->n_pred && best_prev < 0; ++k)
        {
            const pgo_node* cn = &nodes[cur->pred[k]];
            if (cn->len == 0)
                continue; /* gssw would read outside its matrices; not reachable from graphFromJson */
            const int32_t* Hl = cn->H + (size_t)(cn->len - 1) * (size_t)L;
            const int32_t* El = cn->E + (size_t)(cn->len - 1) * (size_t)L;
            if (!gapInRead)
            {
                if (readEnd < 1)
                    continue; /* unreachable: H(0,0) is always explained inside the node */
                int32_t s = sub_score(cur->code[refEnd], q[readEnd]);
                if (score == Hl[readEnd - 1] + s)
                {
                    cig_push_front(&nc->c, match_op(cur->seq[refEnd], read[readEnd]), 1);
                    score -= s;
                    --readEnd;
                    best_prev = cur->pred[k];
                }
            }
            else
            {
                if (score == Hl[readEnd] - GAP_OPEN)
                {
                    cig_push_front(&nc->c, 'D', 1);
                    score += GAP_OPEN;
                    gapInRead = 0;
                    best_prev = cur->pred[k];
                }
                else if (score == El[readEnd] - GAP_EXT)
                {
                    cig_push_front(&nc->c, 'D', 1);
                    score += GAP_EXT;
                    best_prev = cur->pred[k];
                }
            }
        }
        if (best_prev < 0)
        {
            m.status = -1; /* "Could not find a valid previous node" -> reference asserts */
            break;
        }
        n = best_prev;
        refEnd = nodes[n].len - 1;
    }
    /* reverse node order (gssw.c:3526) */
    for (int32_t s = 0, e = m.n_nodes - 1; s < e; ++s, --e)
    {
        node_cigar t = m.nc[s];
        m.nc[s] = m.nc[e];
        m.nc[e] = t;
    }
    m.position = refEnd + 1 < 0 ? 0 : refEnd + 1;
    return m;
}

/* GraphAligner.cpp:170-212 */
static int aligns_end_at_mult_nodes(pgo_graph* g, int dir, int32_t max_node, int32_t L)
{
    pgo_node* nodes = g->nodes[dir];
    int32_t top = nodes[max_node].score1;
    int hits = 0;
    for (uint32_t id = 0; id < g->n_nodes; ++id)
    {
        const pgo_node* n = &nodes[id];
        int found = 0;
        size_t cells = (size_t)n->len * (size_t)L;
        if (top >= 251)
        {
            /* The fill overflowed gssw's byte mode (score + bias >= 255, gssw.c:380, 4100-4104) and was redone in the
             * 16-bit word mode, but alignsEndAtMultNodes still scans the matrix through a uint8_t* for len * readLen
             * BYTES (GraphAligner.cpp:180-187): it sees the low/high bytes of the first half of the uint16 cells. */
            for (size_t b = 0; b < cells; ++b)
            {
                const int32_t v = n->H[b / 2];
                const int32_t byte = (b & 1) ? ((v >> 8) & 0xFF) : (v & 0xFF);
                if (byte == top)
                {
                    found = 1;
                    break;
                }
            }
        }
        else
            for (size_t c = 0; c < cells; ++c)
                if (n->H[c] == top)
                {
                    found = 1;
                    break;
                }
        hits += found;
        if (hits > 1)
            return 1;
    }
    return 0;
}

static int render_cigar(const mapping* m, char* buf, int cap)
{ /* GraphAligner.cpp:88-108 */
    int len = 0;
    char tmp[64];
    for (int32_t i = 0; i < m->n_nodes; ++i)
    {
        int k = snprintf(tmp, sizeof tmp, "%d[", m->nc[i].node);
        if (buf && len + k < cap)
            memcpy(buf + len, tmp, (size_t)k);
        len += k;
        for (int32_t j = 0; j < m->nc[i].c.n; ++j)
        {
            k = snprintf(tmp, sizeof tmp, "%u%c", m->nc[i].c.el[j].length, m->nc[i].c.el[j].type);
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

/* GraphAligner.cpp:214-227: upper-case, fill, traceback, multi flag */
static mapping align_string(pgo_graph* g, int dir, const char* str_in, int len, int* multi, int32_t* max_node_out)
{
    ensure_capacity(g, len);
    char* str = (char*)malloc((size_t)len + 1);
    int8_t* q = (int8_t*)malloc((size_t)len + 1);
    for (int i = 0; i < len; ++i)
    {
        str[i] = (char)toupper((unsigned char)str_in[i]);
        q[i] = nt_code(str[i]);
    }
    str[len] = 0;
    int32_t max_node = fill_graph(g, dir, q, len);
    mapping m = graph_trace_back(g, dir, max_node, str, q, len);
    *multi = aligns_end_at_mult_nodes(g, dir, max_node, len);
    if (max_node_out)
        *max_node_out = max_node;
    free(str);
    free(q);
    return m;
}

int pgo_fill(pgo_graph* g, int dir, const char* str, int len, pgo_fill_result* out, char* cigar_buf, int cigar_cap)
{
    int multi = 0;
    int32_t mn = 0;
    mapping m = align_string(g, dir, str, len, &multi, &mn);
    out->score = m.score;
    out->position = m.position;
    out->max_node = mn;
    out->ref_end = g->nodes[dir][mn].ref_end1;
    out->read_end = g->nodes[dir][mn].read_end1;
    out->multi = multi;
    out->cigar_len = render_cigar(&m, cigar_buf, cigar_cap);
    int st = m.status;
    mapping_free(&m);
    return st;
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

/* GraphAligner::alignRead, GraphAligner.cpp:308-404 */
int pgo_align_read(
    pgo_graph* g, const char* bases, int len, unsigned flags, pgo_result* out, char* cigar_buf, int cigar_cap)
{
    char* rc = (char*)malloc((size_t)len + 1);
    char* rev = (char*)malloc((size_t)len + 1);
    char* rcrev = (char*)malloc((size_t)len + 1);
    for (int i = 0; i < len; ++i)
    {
        rc[i] = complement_base(bases[len - 1 - i]);
        rev[i] = bases[len - 1 - i];
    }
    for (int i = 0; i < len; ++i)
        rcrev[i] = complement_base(rev[len - 1 - i]);
    rc[len] = rev[len] = rcrev[len] = 0;

    mapping gm[4];
    int have[4] = { 1, 0, 0, 0 };
    int multi[4] = { 0, 0, 0, 0 };
    memset(gm, 0, sizeof gm);
    gm[0] = align_string(g, 0, bases, len, &multi[0], NULL);
    if (flags & PGO_AF_BOTH_STRANDS)
    {
        gm[1] = align_string(g, 0, rc, len, &multi[1], NULL);
        have[1] = 1;
    }
    if (flags & PGO_AF_REVERSE_GRAPH)
    {
        gm[2] = align_string(g, 1, rev, len, &multi[2], NULL);
        have[2] = 1;
        if (flags & PGO_AF_BOTH_STRANDS)
        {
            gm[3] = align_string(g, 1, rcrev, len, &multi[3], NULL);
            have[3] = 1;
        }
    }
    int fwd_unique = !multi[0] && !multi[2];
    int rev_unique = !multi[1] && !multi[3];
    int return_reverse = 0;
    if (!fwd_unique && rev_unique && have[1])
        return_reverse = 1;
    else if (fwd_unique && !rev_unique)
        return_reverse = 0;
    else if (have[1])
        return_reverse = gm[0].score < gm[1].score;

    mapping* chosen = return_reverse ? &gm[1] : &gm[0];
    int unique = return_reverse ? rev_unique : fwd_unique;
    out->graph_pos = chosen->position;
    out->score = chosen->score;
    out->unique = unique;
    out->mapq = unique ? 60 : 0;
    out->returned_reverse = return_reverse;
    int status = 0;
    for (int k = 0; k < 4; ++k)
    {
        out->multi[k] = multi[k];
        out->scores[k] = have[k] ? gm[k].score : -1;
        if (have[k] && gm[k].status != 0 && (k < 2))
            status = -1;
    }
    out->cigar_len = (flags & PGO_AF_CIGAR) ? render_cigar(chosen, cigar_buf, cigar_cap) : 0;
    if (!(flags & PGO_AF_CIGAR) && cigar_buf && cigar_cap > 0)
        cigar_buf[0] = 0;
    for (int k = 0; k < 4; ++k)
        if (have[k])
            mapping_free(&gm[k]);
    free(rc);
    free(rev);
    free(rcrev);
    return status;
}

/* ---- threaded batch: the reference's chunking scheme (Align.cpp:114-156) ------------------------ */
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
    pgo_result* results;
    char* cigars;
    int cigar_stride;
    int status;
} pgo_chunk;

static void* chunk_main(void* p)
{
    pgo_chunk* c = (pgo_chunk*)p;
    pgo_graph* g = pgo_graph_create(c->n_nodes, c->seq_off, c->seq, c->pred_off, c->pred);
    for (uint32_t r = c->begin; r < c->end; ++r)
    {
        int len = (int)(c->base_off[r + 1] - c->base_off[r]);
        char* cb = c->cigars ? c->cigars + (size_t)r * (size_t)c->cigar_stride : NULL;
        if (pgo_align_read(g, c->bases + c->base_off[r], len, c->flags, &c->results[r], cb, c->cigars ? c->cigar_stride : 0)
            != 0)
            c->status = -1;
    }
    pgo_graph_destroy(g);
    return NULL;
}

int pgo_align_batch(
    uint32_t n_nodes, const uint32_t* seq_off, const char* seq, const uint32_t* pred_off, const uint32_t* pred,
    uint32_t n_reads, const uint32_t* base_off, const char* bases, unsigned flags, uint32_t threads,
    pgo_result* results, char* cigars, int cigar_stride)
{
    if (threads < 1)
        threads = 1;
    if (threads > n_reads && n_reads > 0)
        threads = n_reads;
    uint32_t step = n_reads ? (n_reads + threads - 1) / threads : 1;
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    pgo_chunk* ch = (pgo_chunk*)calloc(threads, sizeof(pgo_chunk));
    uint32_t used = 0;
    for (uint32_t t = 0; t < threads; ++t)
    {
        uint32_t b = t * step, e = b + step > n_reads ? n_reads : b + step;
        if (b >= e)
            break;
        ch[t] = (pgo_chunk){ n_nodes, seq_off, seq,   pred_off, pred,   base_off,     bases,
                             b,       e,       flags, results,  cigars, cigar_stride, 0 };
        pthread_create(&th[t], NULL, chunk_main, &ch[t]);
        ++used;
    }
    int status = 0;
    for (uint32_t t = 0; t < used; ++t)
    {
        pthread_join(th[t], NULL);
        if (ch[t].status != 0)
            status = -1;
    }
    free(th);
    free(ch);
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * KlibAligner stage -- scalar restatement of klib's ksw_align(KSW_XSTART) + ksw_global, under klib_glue.h.
 *
 *   ksw_qinit (16-bit profile, slen = ceil(qlen/8))   external/klib/ksw.c:58-104
 *   ksw_i16                                           external/klib/ksw.c:223-321
 *   ksw_align (reverse pass with KSW_XSTOP)           external/klib/ksw.c:330-355
 *   ksw_global (+ push_cigar run merging)             external/klib/ksw.c:441-455, 457-531
 *
 * ksw_i16 is Farrar's striped kernel.  Written cell by cell it is the textbook affine recurrence EXCEPT for one
 * thing this restatement keeps: E of the next column is taken from the H of the FIRST pass over the column, i.e. with
 * the vertical gap F restarted at every stripe segment (rows that are multiples of slen); the lazy-F loop later raises
 * H but never E (ksw.c:262-288).  The column maximum / end-row rule is the memory order of the striped vector
 * (ksw.c:304-306): smallest row % slen first, then smallest row / slen.  Padding rows (>= qlen) can never hold a new
 * maximum and are not modelled.
 * ---------------------------------------------------------------------------------------------- */
#include "klib_glue.h"

static inline int pgo_sat0(int v) { return v < 0 ? 0 : v; }

/* target accessor of ksw_align's second pass: the first te+1 columns are reversed in place, the rest are not */
static inline uint8_t pgo_tcol(const uint8_t* t, int i, int rev_te) { return (rev_te >= 0 && i <= rev_te) ? t[rev_te - i] : t[i]; }

/* one ksw_i16 call: q[0..qlen) (already in the orientation of the pass), endsc = 0x10000 for "no KSW_XSTOP" */
static void pgo_ksw_i16(
    int qlen, const uint8_t* q, int tlen, const uint8_t* t, int rev_te, const int8_t* mat, int gapo, int gape, int endsc, int* score,
    int* te_out, int* qe_out)
{
    const int slen = (qlen + 7) / 8, gapoe = gapo + gape;
    int* H0 = (int*)calloc((size_t)qlen + 1, sizeof(int));
    int* H1 = (int*)calloc((size_t)qlen + 1, sizeof(int));
    int* E = (int*)calloc((size_t)qlen + 1, sizeof(int));
    int* Hmax = (int*)calloc((size_t)qlen + 1, sizeof(int));
    int gmax = 0, te = -1;
    for (int i = 0; i < tlen; ++i)
    {
        const int8_t* row = mat + 5 * pgo_tcol(t, i, rev_te);
        int f_loc = 0, f_full = 0, diag = 0, imax = 0;
        for (int j = 0; j < qlen; ++j)
        {
            if (j % slen == 0)
                f_loc = 0;
            int base = diag + row[q[j]];
            if (base < E[j])
                base = E[j];
            const int hp = base > f_loc ? base : f_loc;
            const int h = base > f_full ? base : f_full;
            diag = H0[j];
            H1[j] = h;
            if (h > imax)
                imax = h;
            int e = pgo_sat0(E[j] - gape), o = pgo_sat0(hp - gapoe);
            E[j] = e > o ? e : o;
            f_loc = pgo_sat0(f_loc - gape);
            if (f_loc < o)
                f_loc = o;
            f_full = pgo_sat0(f_full - gape);
            const int of = pgo_sat0(h - gapoe);
            if (f_full < of)
                f_full = of;
        }
        if (imax > gmax)
        {
            gmax = imax;
            te = i;
            memcpy(Hmax, H1, (size_t)qlen * sizeof(int));
            if (gmax >= endsc)
                break;
        }
        int* s = H1;
        H1 = H0;
        H0 = s;
    }
    int best = -1, qe = -1;
    long best_key = 0;
    for (int j = 0; j < qlen; ++j)
    {
        const long key = (long)(j % slen) * 8 + j / slen;
        if (Hmax[j] > best || (Hmax[j] == best && key < best_key))
        {
            best = Hmax[j];
            best_key = key;
            qe = j;
        }
    }
    *score = gmax;
    *te_out = te;
    *qe_out = qe;
    free(H0);
    free(H1);
    free(E);
    free(Hmax);
}

#define PGO_MINUS_INF (-0x40000000)

static uint32_t* pgo_push_cigar(int* n, int* m, uint32_t* c, int op, int len)
{
    if (*n == 0 || (uint32_t)op != (c[*n - 1] & 0xf))
    {
        if (*n == *m)
        {
            *m = *m ? *m << 1 : 4;
            c = (uint32_t*)realloc(c, (size_t)*m * 4);
        }
        c[(*n)++] = (uint32_t)len << 4 | (uint32_t)op;
    }
    else
        c[*n - 1] += (uint32_t)len << 4;
    return c;
}

static void pgo_ksw_global(int qlen, const uint8_t* q, int tlen, const uint8_t* t, const int8_t* mat, int gapo, int gape, int w, int* n_cigar_, uint32_t** cigar_)
{
    const int gapoe = gapo + gape;
    const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    uint8_t* z = (uint8_t*)malloc((size_t)(n_col > 0 ? n_col : 1) * (size_t)(tlen > 0 ? tlen : 1));
    int* eh_h = (int*)calloc((size_t)qlen + 2, sizeof(int));
    int* eh_e = (int*)calloc((size_t)qlen + 2, sizeof(int));
    eh_h[0] = 0;
    eh_e[0] = PGO_MINUS_INF;
    int j;
    for (j = 1; j <= qlen && j <= w; ++j)
    {
        eh_h[j] = -(gapo + gape * j);
        eh_e[j] = PGO_MINUS_INF;
    }
    for (; j <= qlen; ++j)
        eh_h[j] = eh_e[j] = PGO_MINUS_INF;
    for (int i = 0; i < tlen; ++i)
    {
        int f = PGO_MINUS_INF, h1;
        const int8_t* row = mat + 5 * t[i];
        uint8_t* zi = z + (size_t)i * (size_t)n_col;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(gapo + gape * (i + 1)) : PGO_MINUS_INF;
        for (j = beg; j < end; ++j)
        {
            int h = eh_h[j], e = eh_e[j];
            uint8_t d;
            eh_h[j] = h1;
            h += row[q[j]];
            d = h > e ? 0 : 1;
            h = h > e ? h : e;
            d = h > f ? d : 2;
            h = h > f ? h : f;
            h1 = h;
            h -= gapoe;
            e -= gape;
            d |= e > h ? 1 << 2 : 0;
            e = e > h ? e : h;
            eh_e[j] = e;
            f -= gape;
            d |= f > h ? 2 << 4 : 0;
            f = f > h ? f : h;
            zi[j - beg] = d;
        }
        eh_h[end] = h1;
        eh_e[end] = PGO_MINUS_INF;
    }
    int n = 0, m = 0, which = 0;
    uint32_t* cigar = 0;
    int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    while (i >= 0 && k >= 0)
    {
        which = z[(size_t)i * (size_t)n_col + (size_t)(k - (i > w ? i - w : 0))] >> (which << 1) & 3;
        if (which == 0)
        {
            cigar = pgo_push_cigar(&n, &m, cigar, 0, 1);
            --i;
            --k;
        }
        else if (which == 1)
        {
            cigar = pgo_push_cigar(&n, &m, cigar, 2, 1);
            --i;
        }
        else
        {
            cigar = pgo_push_cigar(&n, &m, cigar, 1, 1);
            --k;
        }
    }
    if (i >= 0)
        cigar = pgo_push_cigar(&n, &m, cigar, 2, i + 1);
    if (k >= 0)
        cigar = pgo_push_cigar(&n, &m, cigar, 1, k + 1);
    for (i = 0; i < n >> 1; ++i)
    {
        uint32_t tmp = cigar[i];
        cigar[i] = cigar[n - 1 - i];
        cigar[n - 1 - i] = tmp;
    }
    *n_cigar_ = n;
    *cigar_ = cigar;
    free(eh_h);
    free(eh_e);
    free(z);
}

static void pgo_klib_engine(int qlen, uint8_t* query, int tlen, uint8_t* target, const int8_t* mat, int gapo, int gape, klib_pair* out)
{
    int score, te, qe;
    pgo_ksw_i16(qlen, query, tlen, target, -1, mat, gapo, gape, 0x10000, &score, &te, &qe);
    /* second pass: query[0..qe] reversed against target with its first te+1 columns reversed, stop at >= score */
    uint8_t* rq = (uint8_t*)malloc((size_t)qe + 2);
    for (int j = 0; j <= qe; ++j)
        rq[j] = query[qe - j];
    int rscore, rte, rqe;
    pgo_ksw_i16(qe + 1, rq, tlen, target, te, mat, gapo, gape, score, &rscore, &rte, &rqe);
    free(rq);
    out->score = score;
    out->te = te;
    out->qe = qe;
    out->tb = out->qb = -1;
    if (rscore == score)
    {
        out->tb = te - rte;
        out->qb = qe - rqe;
    }
    out->n_cigar = 0;
    out->cigar = 0;
    out->ub = (out->tb < 0 || out->qb < 0);
    if (out->ub)
        return;
    pgo_ksw_global(qe - out->qb + 1, query + out->qb, te - out->tb + 1, target + out->tb, mat, gapo, gape, tlen, &out->n_cigar, &out->cigar);
}

int pgo_klib_pair(const char* ref, const char* query, int match, int mismatch, int gapo, int gape, int32_t* out5, uint32_t* cigar, int cap)
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
    pgo_klib_engine(ql, q, tl, t, mat, gapo, gape, &pr);
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

int pgo_klib_align(
    int n_nodes, const uint32_t* node_off, const char* node_seq, int n_paths, const uint32_t* path_node_off, const uint32_t* path_nodes,
    uint32_t n_reads, const uint32_t* read_off, const char* read_bases, const uint8_t* bam_reverse, int match, int mismatch, int gapo,
    int gape, klib_result* results, char* cigars, int stride)
{
    return klib_align_batch(
        pgo_klib_engine, n_nodes, node_off, node_seq, n_paths, path_node_off, path_nodes, n_reads, read_off, read_bases, bam_reverse,
        match, mismatch, gapo, gape, results, cigars, stride);
}
