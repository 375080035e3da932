// CPU harness of the general gssw stage's scalar core (paragraph_amd/csrc/pg_general.h): builds the graph tables of ONE
// graph the way pg_graphs_upload lays them out (both directions, predecessors ascending, upper-cased forward characters)
// and runs the four fills + strand pick + traceback on the host.  Test infrastructure: tests/test_general_cpu.py compares it
// with the reference's gssw.c; the product reaches the same code only through pg_general.hip's kernels.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../paragraph_amd/csrc/pg_general.h"

extern "C" int pgt_general_align(
    uint32_t n_nodes, const uint32_t* seq_off, const char* seq, const uint32_t* pred_off, const uint32_t* pred, const char* bases,
    uint32_t L, uint32_t flags, pg_result* result, uint32_t* ops, uint32_t ops_cap, uint32_t* n_ops, int32_t* fill_out /* [4][6] */)
{
    PgGraphDev gd{};
    std::vector<PgNode> nodes;
    std::vector<uint32_t> preds;
    std::vector<char> seqchars;
    std::vector<std::vector<uint32_t>> succ(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i)
        for (uint32_t k = pred_off[i]; k < pred_off[i + 1]; ++k)
            succ[pred[k]].push_back(i);
    uint32_t total = 0;
    for (int dir = 0; dir < 2; ++dir)
    {
        gd.dir[dir].node_off = (uint32_t)nodes.size();
        gd.dir[dir].n_nodes = n_nodes;
        uint32_t col = 0;
        for (uint32_t id = 0; id < n_nodes; ++id)
        {
            const uint32_t src = dir ? n_nodes - 1 - id : id;
            const uint32_t s0 = seq_off[src], len = seq_off[src + 1] - s0;
            PgNode nd{};
            nd.col_start = col;
            nd.len = len;
            nd.pred_off = (uint32_t)preds.size();
            if (!dir)
            {
                for (uint32_t k = pred_off[id]; k < pred_off[id + 1]; ++k)
                    preds.push_back(pred[k]);
                for (uint32_t c = 0; c < len; ++c)
                {
                    const char ch = seq[s0 + c];
                    seqchars.push_back((ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch);
                }
            }
            else
            {
                std::vector<uint32_t> ps;
                for (uint32_t s : succ[src])
                    ps.push_back(n_nodes - 1 - s);
                std::sort(ps.begin(), ps.end());
                preds.insert(preds.end(), ps.begin(), ps.end());
            }
            nd.n_pred = (uint32_t)preds.size() - nd.pred_off;
            nodes.push_back(nd);
            col += len;
        }
        total = col;
        gd.dir[dir].ncols = col;
    }
    if (preds.empty())
        preds.push_back(0);
    std::vector<int16_t> H((size_t)2 * total * L), seeds((size_t)4 * 2 * n_nodes * L), cols((size_t)4 * 2 * L);
    std::vector<int32_t> node_max((size_t)4 * 2 * n_nodes);
    PgFillSummary fsum[4];
    memset(fsum, 0, sizeof fsum);
    for (int f = 0; f < 4; ++f)
    {
        const int dir = f >> 1, strand = f & 1;
        if ((strand && !(flags & PG_AF_BOTH_STRANDS)) || (dir && !(flags & PG_AF_REVERSE_GRAPH)))
            continue;
        int16_t* sH = seeds.data() + (size_t)f * 2 * n_nodes * L;
        pggen::fill(gd, nodes.data(), preds.data(), seqchars.data(), dir, strand, bases, (int)L,
                    dir == 0 ? H.data() + (size_t)strand * total * L : nullptr, sH, sH + (size_t)n_nodes * L,
                    cols.data() + (size_t)f * 2 * L, cols.data() + (size_t)f * 2 * L + L, node_max.data() + (size_t)f * 2 * n_nodes, &fsum[f]);
        if (fill_out)
        {
            int32_t* o = fill_out + f * 6;
            o[0] = fsum[f].score, o[1] = fsum[f].max_node, o[2] = fsum[f].ref_end, o[3] = fsum[f].read_end, o[4] = fsum[f].end_col, o[5] = fsum[f].multi;
        }
    }
    std::vector<uint32_t> scratch(pg_gen_ops_cap(L));
    const uint32_t n = pggen::pick_and_trace(gd, nodes.data(), preds.data(), seqchars.data(), bases, (int)L, flags, fsum, H.data(), seeds.data(),
                                             scratch.data(), result);
    *n_ops = n;
    if (n > ops_cap)
        return 1;
    for (uint32_t e = 0; e < n; ++e)
        ops[e] = scratch[scratch.size() - n + e];
    result->ops_off = 0;
    return 0;
}
