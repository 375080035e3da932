// Host-side mirror of the reference's grm:: / paragraph:: interfaces over the C ABI of libparagraph_amd.so.
// Everything here is marshalling: graphs -> CSR, reads -> packed bytes, pg_result/pg_op -> common::Read.
// No alignment arithmetic happens on the host and there is no CPU fallback: without a device every call throws.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>

#include "../../../include/paragraph_amd.h"
#include "grm/Align.hh"
#include "grm/CompositeAligner.hh"
#include "grm/GraphAligner.hh"
#include "grm/KmerAligner.hh"
#include "grm/PathAligner.hh"
#include "grm/ValidationAligner.hh"
#include "paragraph/SiteBatcher.hh"
#include "parallel.hh"
#include "pinned.hh"

using common::Read;
using graphtools::Graph;
using graphtools::NodeId;

namespace
{
// PG_ERR_UNSUPPORTED: the input is outside the envelope of the device kernels (not a failure of the device).  SiteBatcher
// catches this one per site, so that one oversize site does not take a run over thousands of sites down.
struct OutsideEnvelope : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

void check(pg_ctx* ctx, pg_status st, const char* what)
{
    if (st == PG_OK)
        return;
    const std::string msg = std::string(what) + ": " + pg_strerror(st) + " (" + (ctx ? pg_last_error(ctx) : "") + ")";
    if (st == PG_ERR_UNSUPPORTED)
        throw OutsideEnvelope(msg);
    throw std::runtime_error(msg);
}

// ---- devices: slot s of the device list -> one pg_ctx (own streams, workspace, stage mutex, batch pool) -----------------
// The list comes from paragraph::setDevices(), else PG_DEVICES ("0,1,2,3" or "all"), else PG_DEVICE, else {0}.  The
// reference's only parallelism is thread-per-chunk / thread-per-(sample, graph) (Align.cpp:114-156, grmpy/Workflow.cpp:
// 225-231); the device analogue is lane-per-chunk with the lanes spread over the devices, no data-path collective.
struct DeviceSlot
{
    int ordinal = 0;
    pg_ctx* ctx = nullptr;
    std::mutex create;       // guards ctx creation
    std::mutex stage;        // the stage calls on one ctx must be serialised (paragraph_amd.h)
    std::mutex pool_mutex;
    std::vector<pg_batch*> idle;
};
struct DeviceTable
{
    std::mutex m;
    std::vector<std::unique_ptr<DeviceSlot>> slots;
    bool fixed = false;  // a context exists: the list can no longer change
};
DeviceTable& deviceTable()
{
    static DeviceTable* t = new DeviceTable();  // never torn down (like the HIP runtime itself)
    return *t;
}
std::vector<int> devicesFromEnvironment()
{
    std::vector<int> out;
    if (const char* list = std::getenv("PG_DEVICES"))
    {
        const std::string s(list);
        if (s == "all")
        {
            // pg_ctx_create fails with PG_ERR_NO_DEVICE past the last ordinal
            for (int d = 0; d < 64; ++d)
            {
                pg_ctx* probe = nullptr;
                if (!std::getenv("PG_SPIN_WAITS"))
                    (void)pg_device_prefer_blocking_waits(d);  // before the device's first use (see deviceContext)
                if (pg_ctx_create(d, &probe) != PG_OK)
                    break;
                pg_ctx_destroy(probe);
                out.push_back(d);
            }
        }
        else
        {
            size_t at = 0;
            while (at < s.size())
            {
                size_t end = s.find(',', at);
                if (end == std::string::npos)
                    end = s.size();
                if (end > at)
                    out.push_back(std::atoi(s.substr(at, end - at).c_str()));
                at = end + 1;
            }
        }
    }
    if (out.empty())
    {
        const char* dev = std::getenv("PG_DEVICE");
        out.push_back(dev ? std::atoi(dev) : 0);
    }
    return out;
}
void ensureDeviceList(DeviceTable& t)
{
    if (t.slots.empty())
        for (int d : devicesFromEnvironment())
        {
            t.slots.emplace_back(new DeviceSlot());
            t.slots.back()->ordinal = d;
        }
}
DeviceSlot& deviceSlot(int slot)
{
    DeviceTable& t = deviceTable();
    std::lock_guard<std::mutex> lock(t.m);
    ensureDeviceList(t);
    if (slot < 0 || (size_t)slot >= t.slots.size())
        throw std::out_of_range("device slot " + std::to_string(slot) + " of " + std::to_string(t.slots.size()));
    return *t.slots[(size_t)slot];
}

pg_ctx* deviceContext(int slot = 0)
{
    DeviceSlot& ds = deviceSlot(slot);
    std::lock_guard<std::mutex> lock(ds.create);
    if (!ds.ctx)
    {
        // lanes wait for their batches while other lanes need the CPU (PG_SPIN_WAITS keeps HIP's spinning default)
        if (!std::getenv("PG_SPIN_WAITS"))
            (void)pg_device_prefer_blocking_waits(ds.ordinal);
        pg_status st = pg_ctx_create(ds.ordinal, &ds.ctx);
        if (st != PG_OK)
            throw std::runtime_error("pg_ctx_create(device " + std::to_string(ds.ordinal) + "): " + pg_strerror(st));
        {
            std::lock_guard<std::mutex> tl(deviceTable().m);
            deviceTable().fixed = true;
        }
        // Workspace budget (H trace + seeds of the reads in flight; allocated on demand up to this): the library's default
        // of 8 GiB cuts a 1000-site batch into ~9 chunks of 25 k reads, too few threads for the one-thread-per-read
        // traceback kernel.  An MI355X has 288 GB: 64 GiB (what bench.py uses) keeps such a batch in one or two chunks.
        // (pg_ctx_set_fill_streams(ctx, 2) -- fills alternating over two streams and three workspace regions, so that a launch's
        // draining tail is filled by the next chunk -- was measured here: with the two streams on two hardware queues
        // (PG_FILLS_LOW=1) the fills do overlap, and this workflow loses 3.5 %: two batches that share the chip both finish late,
        // and a lane waiting for its batch prepares nothing (profiles/r05_fill_streams_queues_ab.jsonl).  PG_FILL_STREAMS=2
        // PG_FILLS_LOW=1 in the environment switches it on.)
        const char* gib = std::getenv("PG_WORKSPACE_GIB");
        const double budget_gib = gib ? std::atof(gib) : 64.0;
        if (budget_gib > 0)
        {
            st = pg_ctx_set_workspace_bytes(ds.ctx, (uint64_t)(budget_gib * (double)(1ull << 30)));
            if (st != PG_OK)
                throw std::runtime_error(std::string("pg_ctx_set_workspace_bytes: ") + pg_strerror(st));
        }
    }
    return ds.ctx;
}
std::mutex& deviceMutex(int slot = 0) { return deviceSlot(slot).stage; }

// Batch objects keep their device buffers between uses: SiteBatcher::run takes one from here and hands it back, so
// that after the first few batches no run() allocates or frees device memory (hipMalloc / hipFree stall the queues).
struct BatchPool
{
    DeviceSlot& ds;
    pg_batch* take(pg_ctx* ctx)
    {
        {
            std::lock_guard<std::mutex> lock(ds.pool_mutex);
            if (!ds.idle.empty())
            {
                pg_batch* b = ds.idle.back();
                ds.idle.pop_back();
                return b;
            }
        }
        pg_batch* b = nullptr;
        check(ctx, pg_batch_create(ctx, &b), "pg_batch_create");
        return b;
    }
    void giveBack(pg_ctx* ctx, pg_batch* b, bool reusable)
    {
        static const size_t keep = 64;  // (a workflow's lanes keep two batches in flight each)
        if (reusable)
        {
            std::lock_guard<std::mutex> lock(ds.pool_mutex);
            if (ds.idle.size() < keep)
            {
                ds.idle.push_back(b);
                return;
            }
        }
        pg_batch_destroy(ctx, b);
    }
};
BatchPool batchPool(int slot = 0) { return BatchPool{ deviceSlot(slot) }; }

struct GraphCsr
{
    std::vector<uint32_t> node_off{ 0 }, seq_off{ 0 }, pred_off{ 0 }, pred, n_labels;
    // label sets of the edges, in predecessor-entry order: bit sets of `label_words` 64-bit words each (one word unless a graph
    // of the set has more than 64 labels: pg_graphs_set_labels_wide).  Kept as lists of bit indices until the set is complete.
    std::vector<std::vector<uint16_t>> label_bits;
    uint32_t label_words = 1;
    std::vector<uint64_t> label_mask;  // [pred entry][label_words], made by finish()
    std::string seq;
    std::vector<std::vector<std::string>> label_names;  // per graph, sorted
    void finish()
    {
        label_mask.assign(label_bits.size() * (size_t)label_words, 0);
        for (size_t q = 0; q < label_bits.size(); ++q)
            for (uint16_t b : label_bits[q])
                label_mask[q * label_words + (b >> 6)] |= 1ull << (b & 63);
    }
    void add(const Graph& g)
    {
        std::vector<std::string> names;
        {
            auto all = g.allLabels();
            names.assign(all.begin(), all.end());
        }
        if (names.size() > PG_MAX_LABELS)
            throw OutsideEnvelope("more than 256 sequence labels on one graph (paragraph_amd.h: PG_MAX_LABELS)");
        label_words = std::max<uint32_t>(label_words, (uint32_t)((names.size() + 63) / 64));
        for (NodeId n = 0; n != g.numNodes(); ++n)
        {
            const std::string& s = g.nodeSeq(n);
            for (char c : s)
            {
                const char u = (char)std::toupper((unsigned char)c);
                if (g.isSequenceExpansionRequired() && n != 0 && n != g.numNodes() - 1 && u != 'A' && u != 'C' && u != 'G'
                    && u != 'T' && u != 'X')
                    throw OutsideEnvelope("degenerate node sequences need node expansion, which the device path does not do");
            }
            seq += s;
            seq_off.push_back((uint32_t)seq.size());
            for (NodeId p : g.predecessors(n))
            {
                pred.push_back(p);
                std::vector<uint16_t> bits;
                for (auto const& l : g.edgeLabels(p, n))
                    bits.push_back((uint16_t)(std::lower_bound(names.begin(), names.end(), l) - names.begin()));
                label_bits.push_back(std::move(bits));
            }
            pred_off.push_back((uint32_t)pred.size());
        }
        node_off.push_back(node_off.back() + (uint32_t)g.numNodes());
        n_labels.push_back((uint32_t)names.size());
        label_names.push_back(names);
    }
};

std::string reverseComplement(std::string s)
{  // GT!/src/graphutils/SequenceOperations.cpp:66-88
    for (char& c : s)
        c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
    std::reverse(s.begin(), s.end());
    return s;
}

// GraphAligner.cpp:358-401
// reverse_quals: only GraphAligner reverses the qualities with the bases (GraphAligner.cpp:375-378); the k-mer and klib
// stages replace the bases alone (KmerAligner.cpp:457, KlibAligner.cpp:328)
void applyResult(Read& read, const pg_result& r, const pg_op* ops, bool want_cigar, bool reverse_quals = true)
{
    if (r.returned_reverse)
    {
        read.set_bases(reverseComplement(read.bases()));
        if (reverse_quals)
        {
            std::string q = read.quals();
            std::reverse(q.begin(), q.end());
            read.set_quals(q);
        }
    }
    read.set_is_graph_reverse_strand(read.is_reverse_strand() != (r.returned_reverse != 0));
    read.set_graph_pos(r.graph_pos);
    read.set_graph_alignment_score(r.score);
    read.set_is_graph_alignment_unique(r.is_unique != 0);
    read.set_graph_mapq(r.mapq);
    if (want_cigar)
    {
        std::string buf(16 + 12 * (size_t)r.n_ops, '\0');
        const size_t n = pg_render_cigar(&r, ops, &buf[0], buf.size());
        buf.resize(n);
        read.set_graph_cigar(buf);
    }
}
}  // namespace

namespace pghost
{
namespace
{
struct PinnedPool
{
    std::mutex m;
    std::multimap<size_t, PinnedBlock> idle;  // by capacity
    size_t idle_bytes = 0, allocated = 0;
    bool warned = false;
};
PinnedPool& pinnedPool()
{
    static PinnedPool* p = new PinnedPool();
    return *p;
}
const size_t kPinnedIdleMax = 4ull << 30;
size_t pinnedClass(size_t bytes)
{
    size_t c = 4096;
    while (c < bytes)
        c <<= 1;
    if (c >= 65536)
    {
        const size_t step = c / 8;  // four classes between c / 2 and c
        c = (bytes + step - 1) / step * step;
    }
    return c;
}
}  // namespace

PinnedBlock pinnedTake(size_t bytes)
{
    PinnedPool& pool = pinnedPool();
    const size_t want = pinnedClass(std::max<size_t>(bytes, 1));
    {
        std::lock_guard<std::mutex> lock(pool.m);
        auto it = pool.idle.lower_bound(want);
        if (it != pool.idle.end() && it->first <= want + want / 2)
        {
            PinnedBlock blk = it->second;
            pool.idle_bytes -= blk.bytes;
            pool.idle.erase(it);
            return blk;
        }
    }
    PinnedBlock blk;
    blk.bytes = want;
    pg_ctx* ctx = deviceContext(0);
    if (pg_host_alloc(ctx, want, &blk.p) == PG_OK)
    {
        blk.pinned = true;
        std::lock_guard<std::mutex> lock(pool.m);
        pool.allocated += want;
        return blk;
    }
    blk.p = std::malloc(want);
    if (!blk.p)
        throw std::bad_alloc();
    std::lock_guard<std::mutex> lock(pool.m);
    if (!pool.warned)
        fprintf(stderr, "paragraph_amd: page-locking %zu bytes of staging memory failed (%s); copies go through pageable memory\n",
                want, pg_last_error(ctx));
    pool.warned = true;
    return blk;
}

void pinnedGive(PinnedBlock const& blk)
{
    if (!blk.p)
        return;
    PinnedPool& pool = pinnedPool();
    {
        std::lock_guard<std::mutex> lock(pool.m);
        if (pool.idle_bytes + blk.bytes <= kPinnedIdleMax)
        {
            pool.idle.emplace(blk.bytes, blk);
            pool.idle_bytes += blk.bytes;
            return;
        }
    }
    if (blk.pinned)
        pg_host_free(deviceContext(0), blk.p);
    else
        std::free(blk.p);
}

size_t pinnedBytesAllocated()
{
    std::lock_guard<std::mutex> lock(pinnedPool().m);
    return pinnedPool().allocated;
}
}  // namespace pghost

namespace paragraph
{
void setDevices(std::vector<int> const& ordinals)
{
    DeviceTable& t = deviceTable();
    std::lock_guard<std::mutex> lock(t.m);
    if (t.fixed)
    {
        std::vector<int> now;
        for (auto const& s : t.slots)
            now.push_back(s->ordinal);
        if (now == ordinals || ordinals.empty())
            return;
        throw std::logic_error("paragraph::setDevices: the device list cannot change once a device context exists");
    }
    if (ordinals.empty())
        return;
    t.slots.clear();
    for (int d : ordinals)
    {
        t.slots.emplace_back(new DeviceSlot());
        t.slots.back()->ordinal = d;
    }
}

size_t deviceCount()
{
    DeviceTable& t = deviceTable();
    std::lock_guard<std::mutex> lock(t.m);
    ensureDeviceList(t);
    return t.slots.size();
}

size_t pinnedStagingBytes() { return pghost::pinnedBytesAllocated(); }
int usableCpus() { return pghost::usableCpus(); }
}  // namespace paragraph

namespace grm
{
struct GraphAligner::GraphAlignerImpl
{
    pg_graphs* graphs = nullptr;
    const Graph* graph = nullptr;
    ~GraphAlignerImpl()
    {
        if (graphs)
            pg_graphs_destroy(deviceContext(), graphs);
    }
};

GraphAligner::GraphAligner() : _impl(new GraphAlignerImpl()) {}
GraphAligner::~GraphAligner() = default;
GraphAligner::GraphAligner(GraphAligner&& rhs) noexcept : _impl(std::move(rhs._impl)) {}
GraphAligner& GraphAligner::operator=(GraphAligner&& rhs) noexcept
{
    _impl = std::move(rhs._impl);
    return *this;
}

void GraphAligner::setGraph(Graph const* g)
{
    pg_ctx* ctx = deviceContext();
    std::lock_guard<std::mutex> lock(deviceMutex());
    if (_impl->graphs)
        pg_graphs_destroy(ctx, _impl->graphs);
    _impl->graphs = nullptr;
    _impl->graph = g;
    GraphCsr csr;
    csr.add(*g);
    check(ctx, pg_graphs_upload(ctx, 1, csr.node_off.data(), csr.seq_off.data(), csr.seq.data(), csr.pred_off.data(),
                                csr.pred.empty() ? nullptr : csr.pred.data(), &_impl->graphs),
          "pg_graphs_upload");
}

void GraphAligner::alignReads(std::vector<Read*> const& reads, unsigned int flags) const
{
    if (!_impl->graphs)
        throw std::logic_error("GraphAligner::setGraph has not been called");
    pg_ctx* ctx = deviceContext();
    std::vector<uint32_t> base_off{ 0 }, gor(reads.size(), 0);
    std::string bases;
    for (Read* r : reads)
    {
        bases += r->bases();
        base_off.push_back((uint32_t)bases.size());
    }
    std::vector<pg_result> res(reads.size());
    std::vector<pg_op> ops(bases.size() + 48 * reads.size() + 1);
    uint64_t n_ops = 0;
    {
        std::lock_guard<std::mutex> lock(deviceMutex());
        check(ctx, pg_align_batch(ctx, _impl->graphs, (uint32_t)reads.size(), gor.data(), base_off.data(), bases.data(),
                                  flags, res.data(), ops.data(), ops.size(), &n_ops),
              "pg_align_batch");
    }
    for (size_t i = 0; i < reads.size(); ++i)
    {
        if (reads[i]->bases().empty())
            continue;
        if (res[i].status == 2)
            throw std::runtime_error("device traceback inconsistency (the reference would assert)");
        applyResult(*reads[i], res[i], ops.data(), (flags & AF_CIGAR) != 0);
    }
}

void GraphAligner::alignRead(Read& read, unsigned int flags) const
{
    std::vector<Read*> one{ &read };
    alignReads(one, flags);
}

std::string GraphAligner::align(const std::string& read, int& mapq, int& position, int& score) const
{
    Read tmp;
    tmp.set_bases(read);
    alignRead(tmp, AF_CIGAR);  // GraphAligner.cpp:287-296
    mapq = tmp.graph_mapq();
    position = tmp.graph_pos();
    score = tmp.graph_alignment_score();
    return tmp.graph_cigar();
}

struct PathAligner::Impl
{
    int32_t kmer_size = 32;
    pg_graphs* graphs = nullptr;
    ~Impl()
    {
        if (graphs)
            pg_graphs_destroy(deviceContext(), graphs);
    }
};

PathAligner::PathAligner(int32_t kmer_size) : impl_(new Impl()) { impl_->kmer_size = kmer_size; }
PathAligner::~PathAligner() = default;
PathAligner::PathAligner(PathAligner&& rhs) noexcept = default;
PathAligner& PathAligner::operator=(PathAligner&& rhs) noexcept = default;

void PathAligner::setGraph(Graph const* g, std::list<graphtools::Path> const&)
{
    pg_ctx* ctx = deviceContext();
    std::lock_guard<std::mutex> lock(deviceMutex());
    if (impl_->graphs)
        pg_graphs_destroy(ctx, impl_->graphs);
    impl_->graphs = nullptr;
    GraphCsr csr;
    csr.add(*g);
    check(ctx, pg_graphs_upload(ctx, 1, csr.node_off.data(), csr.seq_off.data(), csr.seq.data(), csr.pred_off.data(),
                                csr.pred.empty() ? nullptr : csr.pred.data(), &impl_->graphs),
          "pg_graphs_upload");
    check(ctx, pg_graphs_build_path_index(ctx, impl_->graphs, (uint32_t)impl_->kmer_size), "pg_graphs_build_path_index");
}

void PathAligner::alignReads(std::vector<Read*> const& reads)
{
    if (!impl_->graphs)
        throw std::logic_error("PathAligner::setGraph has not been called");
    attempted_ += (unsigned)reads.size();
    if (reads.empty())
        return;
    pg_ctx* ctx = deviceContext();
    std::vector<uint32_t> base_off{ 0 }, gor(reads.size(), 0);
    std::string bases;
    for (Read* r : reads)
    {
        bases += r->bases();
        base_off.push_back((uint32_t)bases.size());
    }
    std::vector<pg_result> res(reads.size());
    std::vector<pg_op> ops(bases.size() + 48 * reads.size() + 1);
    std::vector<uint8_t> flags(reads.size());
    uint64_t n_ops = 0;
    {
        std::lock_guard<std::mutex> lock(deviceMutex());
        pg_batch* b = nullptr;
        check(ctx, pg_batch_create(ctx, &b), "pg_batch_create");
        pg_status st = pg_batch_upload(ctx, b, impl_->graphs, (uint32_t)reads.size(), gor.data(), base_off.data(), bases.data());
        if (st == PG_OK)
            st = pg_batch_path_align(ctx, b);
        if (st == PG_OK)
            st = pg_batch_download_path_flags(ctx, b, flags.data());
        if (st == PG_OK)
            st = pg_batch_download(ctx, b, res.data(), ops.data(), ops.size(), &n_ops);
        pg_batch_destroy(ctx, b);
        check(ctx, st, "path stage");
    }
    for (size_t i = 0; i < reads.size(); ++i)
    {
        if (flags[i] & 2)
            ++anchored_;
        if (!(flags[i] & 1))
            continue;
        Read& read = *reads[i];
        const pg_result& r = res[i];
        // PathAligner.cpp:121-161
        if (r.returned_reverse)
            read.set_bases(reverseComplement(read.bases()));
        read.set_is_graph_reverse_strand(r.returned_reverse != 0);
        std::string buf(16 + 12 * (size_t)r.n_ops, '\0');
        buf.resize(pg_render_cigar(&r, ops.data(), &buf[0], buf.size()));
        read.set_graph_alignment_score(r.score);
        read.set_graph_cigar(buf);
        read.set_graph_pos(r.graph_pos);
        read.set_graph_mapping_status(Read::MAPPED);
        read.set_is_graph_alignment_unique(r.is_unique != 0);
        read.set_graph_mapq(r.mapq);
        ++mapped_;
    }
}

void PathAligner::alignRead(Read& read)
{
    std::vector<Read*> one{ &read };
    alignReads(one);
}

struct KmerAlignerBase::Impl
{
    unsigned k = 16;
    pg_graphs* graphs = nullptr;
    ~Impl()
    {
        if (graphs)
            pg_graphs_destroy(deviceContext(), graphs);
    }
};

KmerAlignerBase::KmerAlignerBase(unsigned kmer_length) : impl_(new Impl()) { impl_->k = kmer_length; }
KmerAlignerBase::~KmerAlignerBase() = default;
KmerAlignerBase::KmerAlignerBase(KmerAlignerBase&& rhs) noexcept = default;
KmerAlignerBase& KmerAlignerBase::operator=(KmerAlignerBase&& rhs) noexcept = default;

void KmerAlignerBase::setGraph(Graph const* g, std::list<graphtools::Path> const& paths)
{
    pg_ctx* ctx = deviceContext();
    std::lock_guard<std::mutex> lock(deviceMutex());
    if (impl_->graphs)
        pg_graphs_destroy(ctx, impl_->graphs);
    impl_->graphs = nullptr;
    GraphCsr csr;
    csr.add(*g);
    check(ctx, pg_graphs_upload(ctx, 1, csr.node_off.data(), csr.seq_off.data(), csr.seq.data(), csr.pred_off.data(),
                                csr.pred.empty() ? nullptr : csr.pred.data(), &impl_->graphs),
          "pg_graphs_upload");
    std::vector<uint32_t> path_off{ 0, (uint32_t)paths.size() }, node_off{ 0 }, nodes;
    for (auto const& p : paths)
    {
        nodes.insert(nodes.end(), p.nodes.begin(), p.nodes.end());
        node_off.push_back((uint32_t)nodes.size());
    }
    if (nodes.empty())
        nodes.push_back(0);
    check(ctx, pg_graphs_build_kmer_index(ctx, impl_->graphs, impl_->k, path_off.data(), node_off.data(), nodes.data()),
          "pg_graphs_build_kmer_index");
}

void KmerAlignerBase::alignReads(std::vector<Read*> const& reads)
{
    if (!impl_->graphs)
        throw std::logic_error("KmerAligner::setGraph has not been called");
    attempted_ += (unsigned)reads.size();
    if (reads.empty())
        return;
    pg_ctx* ctx = deviceContext();
    std::vector<uint32_t> base_off{ 0 }, gor(reads.size(), 0);
    std::string bases;
    for (Read* r : reads)
    {
        bases += r->bases();
        base_off.push_back((uint32_t)bases.size());
        r->set_graph_mapping_status(Read::UNMAPPED);  // KmerAligner.cpp:519
    }
    std::vector<pg_result> res(reads.size());
    std::vector<pg_op> ops(bases.size() + 48 * reads.size() + 1);
    std::vector<uint8_t> flags(reads.size());
    uint64_t n_ops = 0;
    {
        std::lock_guard<std::mutex> lock(deviceMutex());
        pg_batch* b = nullptr;
        check(ctx, pg_batch_create(ctx, &b), "pg_batch_create");
        pg_status st = pg_batch_upload(ctx, b, impl_->graphs, (uint32_t)reads.size(), gor.data(), base_off.data(), bases.data());
        if (st == PG_OK)
            st = pg_batch_kmer_align(ctx, b, PG_AF_ALL);
        if (st == PG_OK)
            st = pg_batch_download_path_flags(ctx, b, flags.data());
        if (st == PG_OK)
            st = pg_batch_download(ctx, b, res.data(), ops.data(), ops.size(), &n_ops);
        pg_batch_destroy(ctx, b);
        check(ctx, st, "k-mer stage");
    }
    for (size_t i = 0; i < reads.size(); ++i)
    {
        if (!(flags[i] & 5))
            continue;
        Read& read = *reads[i];
        // KmerAligner.cpp:424-472 (updateAlignment) + 497-505
        applyResult(read, res[i], ops.data(), true, false);
        read.set_graph_mapq(res[i].mapq);
        read.set_is_graph_alignment_unique(res[i].is_unique != 0);
        read.set_graph_mapping_status((flags[i] & 1) ? Read::MAPPED : Read::BAD_ALIGN);
        mapped_ += (flags[i] & 1) != 0;
    }
}

void KmerAlignerBase::alignRead(Read& read)
{
    std::vector<Read*> one{ &read };
    alignReads(one);
}

struct KlibAligner::Impl
{
    pg_graphs* graphs = nullptr;
    ~Impl()
    {
        if (graphs)
            pg_graphs_destroy(deviceContext(), graphs);
    }
};

KlibAligner::KlibAligner() : impl_(new Impl()) {}
KlibAligner::~KlibAligner() = default;
KlibAligner::KlibAligner(KlibAligner&& rhs) noexcept = default;
KlibAligner& KlibAligner::operator=(KlibAligner&& rhs) noexcept = default;

void KlibAligner::setGraph(Graph const* g, std::list<graphtools::Path> const& paths)
{
    pg_ctx* ctx = deviceContext();
    std::lock_guard<std::mutex> lock(deviceMutex());
    if (impl_->graphs)
        pg_graphs_destroy(ctx, impl_->graphs);
    impl_->graphs = nullptr;
    GraphCsr csr;
    csr.add(*g);
    check(ctx, pg_graphs_upload(ctx, 1, csr.node_off.data(), csr.seq_off.data(), csr.seq.data(), csr.pred_off.data(),
                                csr.pred.empty() ? nullptr : csr.pred.data(), &impl_->graphs),
          "pg_graphs_upload");
    std::vector<uint32_t> path_off{ 0, (uint32_t)paths.size() }, node_off{ 0 }, nodes;
    for (auto const& p : paths)
    {
        nodes.insert(nodes.end(), p.nodes.begin(), p.nodes.end());
        node_off.push_back((uint32_t)nodes.size());
    }
    if (nodes.empty())
        nodes.push_back(0);
    check(ctx, pg_graphs_build_klib_index(ctx, impl_->graphs, path_off.data(), node_off.data(), nodes.data()),
          "pg_graphs_build_klib_index");
}

void KlibAligner::alignReads(std::vector<Read*> const& reads)
{
    if (!impl_->graphs)
        throw std::logic_error("KlibAligner::setGraph has not been called");
    attempted_ += (unsigned)reads.size();
    if (reads.empty())
        return;
    pg_ctx* ctx = deviceContext();
    std::vector<uint32_t> base_off{ 0 }, gor(reads.size(), 0);
    std::string bases;
    for (Read* r : reads)
    {
        bases += r->bases();
        base_off.push_back((uint32_t)bases.size());
        r->set_graph_mapping_status(Read::UNMAPPED);  // KlibAligner.cpp:415
    }
    std::vector<pg_result> res(reads.size());
    std::vector<pg_op> ops(2 * bases.size() + 64 * reads.size() + 1);
    std::vector<uint8_t> flags(reads.size());
    uint64_t n_ops = 0;
    uint32_t overflow = 0;
    {
        std::lock_guard<std::mutex> lock(deviceMutex());
        pg_batch* b = nullptr;
        check(ctx, pg_batch_create(ctx, &b), "pg_batch_create");
        pg_status st = pg_batch_upload(ctx, b, impl_->graphs, (uint32_t)reads.size(), gor.data(), base_off.data(), bases.data());
        if (st == PG_OK)
            st = pg_batch_klib_align(ctx, b, PG_AF_ALL);
        if (st == PG_OK)
            st = pg_batch_download_path_flags(ctx, b, flags.data());
        if (st == PG_OK)
            st = pg_batch_download(ctx, b, res.data(), ops.data(), ops.size(), &n_ops);
        if (st == PG_OK)
            st = pg_graphs_klib_error(ctx, impl_->graphs, &overflow);
        pg_batch_destroy(ctx, b);
        check(ctx, st, "klib stage");
    }
    if (overflow)
        throw std::runtime_error("klib stage: CIGAR buffer overflow on the device");
    for (size_t i = 0; i < reads.size(); ++i)
    {
        if (!(flags[i] & 5))
            continue;
        Read& read = *reads[i];
        // KlibAligner.cpp:310-343 (updateAlignment) + 349-386 (pickBest)
        applyResult(read, res[i], ops.data(), true, false);
        read.set_graph_mapq(res[i].mapq);
        read.set_is_graph_alignment_unique(res[i].is_unique != 0);
        read.set_graph_mapping_status((flags[i] & 1) ? Read::MAPPED : Read::BAD_ALIGN);
        mapped_ += (flags[i] & 1) != 0;
    }
}

void KlibAligner::alignRead(Read& read)
{
    std::vector<Read*> one{ &read };
    alignReads(one);
}

CompositeAligner::CompositeAligner(bool pathMatching, bool graphMatching, bool klibMatching, bool kmerMatching, unsigned flags)
    : on_{ pathMatching, graphMatching, klibMatching, kmerMatching }, gssw_flags_(flags)
{
}
CompositeAligner::~CompositeAligner() = default;
CompositeAligner::CompositeAligner(CompositeAligner&& rhs) noexcept = default;

void CompositeAligner::setGraph(Graph const* graph, std::list<graphtools::Path> const& paths)
{
    if (on_.path)
        path_stage_.setGraph(graph, paths);
    if (on_.graph)
        gssw_stage_.setGraph(graph);
    if (on_.kmer)
        kmer_stage_.setGraph(graph, paths);
    if (on_.klib)
        klib_stage_.setGraph(graph, paths);
}

void CompositeAligner::alignReads(std::vector<Read*> const& all_reads, ReadFilter filter)
{
    tally_.attempted += (unsigned)all_reads.size();
    std::vector<Read*> reads = all_reads;
    if (on_.path)
    {
        // CompositeAligner.cpp:82-103
        const unsigned before = path_stage_.mapped();
        path_stage_.alignReads(reads);
        tally_.path += path_stage_.mapped() - before;
        tally_.anchored = path_stage_.anchored();
        std::vector<Read*> rest;
        for (Read* read : reads)
        {
            if (read->graph_mapping_status() == Read::MAPPED && filter && filter(*read))
            {
                read->set_graph_mapping_status(Read::BAD_ALIGN);
                tally_.filtered += !on_.kmer && !on_.klib && !on_.graph;
            }
            if (read->graph_mapping_status() != Read::MAPPED)
                rest.push_back(read);
        }
        reads.swap(rest);
    }
    if (on_.kmer && !reads.empty())
    {
        // CompositeAligner.cpp:105-126
        kmer_stage_.alignReads(reads);
        std::vector<Read*> rest;
        for (Read* read : reads)
        {
            if (read->graph_mapping_status() == Read::MAPPED)
            {
                if (filter && filter(*read))
                {
                    read->set_graph_mapping_status(Read::BAD_ALIGN);
                    tally_.filtered += !on_.klib && !on_.graph;
                }
                else
                    ++tally_.kmers;
            }
            if (read->graph_mapping_status() != Read::MAPPED)
                rest.push_back(read);
        }
        reads.swap(rest);
    }
    if (on_.klib && !reads.empty())
    {
        // CompositeAligner.cpp:128-150
        klib_stage_.alignReads(reads);
        std::vector<Read*> rest;
        for (Read* read : reads)
        {
            if (read->graph_mapping_status() == Read::MAPPED)
            {
                if (filter && filter(*read))
                {
                    read->set_graph_mapping_status(Read::BAD_ALIGN);
                    tally_.filtered += !on_.graph;
                }
                else
                    ++tally_.klib;
            }
            if (read->graph_mapping_status() != Read::MAPPED)
                rest.push_back(read);
        }
        reads.swap(rest);
    }
    if (!on_.graph || reads.empty())
        return;
    gssw_stage_.alignReads(reads, gssw_flags_);
    for (Read* read : reads)
    {
        // CompositeAligner.cpp:152-175: the gssw stage always produces a mapping
        read->set_graph_mapping_status(Read::MAPPED);
        if (filter && filter(*read))
        {
            read->set_graph_mapping_status(Read::BAD_ALIGN);
            ++tally_.filtered;
        }
        else
            ++tally_.sw;
    }
}

void CompositeAligner::alignRead(Read& read, ReadFilter filter)
{
    std::vector<Read*> one{ &read };
    alignReads(one, filter);
}

void alignReads(
    const Graph* graph, std::list<graphtools::Path> const& paths, std::vector<common::p_Read>& reads, ReadFilter const& filter,
    bool path_sequence_matching, bool graph_sequence_matching, bool klib_sequence_matching, bool kmer_sequence_matching,
    bool validate_alignments, uint32_t /*threads*/)
{
    std::vector<Read*> todo;
    for (auto& r : reads)
    {
        if (r->bases().empty())
            continue;  // Align.cpp:74-77
        r->set_graph_mapping_status(Read::UNMAPPED);
        todo.push_back(r.get());
    }
    if (validate_alignments)
    {  // Align.cpp:96-104
        ValidationAligner<CompositeAligner> aligner(
            CompositeAligner(path_sequence_matching, graph_sequence_matching, klib_sequence_matching, kmer_sequence_matching), graph, paths);
        aligner.setGraph(graph, paths);
        aligner.alignReads(todo, filter);
    }
    else
    {
        CompositeAligner aligner(path_sequence_matching, graph_sequence_matching, klib_sequence_matching, kmer_sequence_matching);
        aligner.setGraph(graph, paths);
        aligner.alignReads(todo, filter);
    }
    std::vector<common::p_Read> kept;
    for (auto& r : reads)
        if (!r->bases().empty() && r->graph_mapping_status() == Read::MAPPED)
            kept.emplace_back(std::move(r));
    reads.swap(kept);  // Align.cpp:155
}
// ---- ValidationAligner (lib/grm/ValidationAligner.cpp) ------------------------------------------------------------------
namespace
{
std::atomic<unsigned> validation_mismapped(0), validation_repeats(0), validation_aligned(0), validation_total(0);
}
template <typename AlignerT>
ValidationAligner<AlignerT>::ValidationAligner(AlignerT&& aligner, const graphtools::Graph* /*graph*/, std::list<graphtools::Path> const& paths)
    : AlignerT(std::move(aligner))
{
    for (auto const& p : paths)
    {
        std::string& nodes = pathNodes_[p.encode()];
        nodes.clear();
        for (auto n : p.nodeIds())
            nodes += (nodes.empty() ? "" : "->") + std::to_string(n);
    }
}
}  // namespace grm

namespace paragraph
{
// the same bookkeeping for the batched workflow (SiteBatcher with BatchParameters::validate_alignments)
void validationAccount(common::Read& read, std::unordered_map<std::string, std::string> const& path_nodes)
{
    ++grm::validation_total;
    if (read.graph_mapping_status() == common::Read::MAPPED)
    {
        ++grm::validation_aligned;
        std::string cigar_nodes;
        bool in_cigar = false;
        for (const char c : read.graph_cigar())
        {
            if (c == '[')
                in_cigar = true;
            else if (c == ']')
                in_cigar = false;
            else if (!in_cigar)
            {
                if (!cigar_nodes.empty())
                    cigar_nodes += "->";
                cigar_nodes += c;
            }
        }
        auto it = path_nodes.find(read.fragment_id().substr(0, read.fragment_id().find('_')));
        const std::string none;
        const bool supports_path = std::string::npos != (it == path_nodes.end() ? none : it->second).find(cigar_nodes);
        grm::validation_mismapped += !supports_path;
        // the reference logs every misplaced read at debug level (ValidationAligner.cpp:84-90); PG_VALIDATE_DEBUG shows the first few
        static const bool debug = std::getenv("PG_VALIDATE_DEBUG") != nullptr;
        static std::atomic<int> shown(0);
        if (debug && !supports_path && shown.fetch_add(1) < 8)
            fprintf(stderr, "misplaced:%s:pn'%s' cn'%s' cigar %s (%zu paths)\n", read.fragment_id().c_str(),
                    it == path_nodes.end() ? "<unknown path>" : it->second.c_str(), cigar_nodes.c_str(), read.graph_cigar().c_str(), path_nodes.size());
    }
    else if (read.graph_mapping_status() == common::Read::BAD_ALIGN && !read.is_graph_alignment_unique())
        ++grm::validation_repeats;
}

std::vector<std::string> validationLogLines()
{
    const unsigned total = grm::validation_total, aligned = grm::validation_aligned, repeats = grm::validation_repeats,
                   mismapped = grm::validation_mismapped;
    char line[160];
    std::vector<std::string> out;
    out.push_back("[VALIDATION]\tMAPQ\tEmpMAPQ\tWrong\tTotal");
    snprintf(line, sizeof line, "[VALIDATION]\tunalgnd\t0\t0\t%u", total - aligned - repeats);
    out.push_back(line);
    snprintf(line, sizeof line, "[VALIDATION]\trepeat\t0\t0\t%u", repeats);
    out.push_back(line);
    const double emp = !mismapped ? 60.0 : aligned ? -10.0 * std::log10((double)mismapped / (double)aligned) : 0.0;
    snprintf(line, sizeof line, "[VALIDATION]\t60\t%g\t%u\t%u", emp, mismapped, aligned);
    out.push_back(line);
    return out;
}
}  // namespace paragraph

namespace grm
{
template <typename AlignerT> void ValidationAligner<AlignerT>::account(Read& read)
{
    ++validation_total;
    if (read.graph_mapping_status() == Read::MAPPED)
    {
        ++validation_aligned;
        const std::string cigarNodes = getNodes(read.graph_cigar());
        const std::string& simulatedPathNodes = pathNodes_[getSimulatedPathId(read)];
        validation_mismapped += std::string::npos == simulatedPathNodes.find(cigarNodes);
    }
    else if (read.graph_mapping_status() == Read::BAD_ALIGN && !read.is_graph_alignment_unique())
        ++validation_repeats;
}
template <typename AlignerT> void ValidationAligner<AlignerT>::alignRead(Read& read, ReadFilter filter)
{
    AlignerT::alignRead(read, filter);
    account(read);
}
template <typename AlignerT> void ValidationAligner<AlignerT>::alignReads(std::vector<Read*> const& reads, ReadFilter filter)
{
    AlignerT::alignReads(reads, filter);
    for (Read* r : reads)
        account(*r);
}
template <typename AlignerT> std::string ValidationAligner<AlignerT>::getNodes(const std::string& cigar)
{
    std::string out;
    bool inCigar = false;
    for (const char c : cigar)
    {
        if (c == '[')
            inCigar = true;
        else if (c == ']')
            inCigar = false;
        else if (!inCigar)
        {
            if (!out.empty())
                out += "->";
            out += c;
        }
    }
    return out;
}
template <typename AlignerT> std::string ValidationAligner<AlignerT>::getSimulatedPathId(Read& read)
{
    return read.fragment_id().substr(0, read.fragment_id().find('_'));
}
template <typename AlignerT> unsigned ValidationAligner<AlignerT>::mismapped() { return validation_mismapped; }
template <typename AlignerT> unsigned ValidationAligner<AlignerT>::repeats() { return validation_repeats; }
template <typename AlignerT> unsigned ValidationAligner<AlignerT>::aligned() { return validation_aligned; }
template <typename AlignerT> unsigned ValidationAligner<AlignerT>::total() { return validation_total; }
template class ValidationAligner<CompositeAligner>;
}  // namespace grm

namespace paragraph
{
void validationAccount(common::Read& read, std::unordered_map<std::string, std::string> const& path_nodes);

struct SiteBatcher::Impl
{
    std::vector<const Graph*> graphs;
    std::vector<std::vector<common::p_Read>*> reads;
    std::vector<const PackedSite*> packed;
    std::vector<std::list<graphtools::Path> const*> paths;
    std::vector<SiteCounts> counts;
    std::vector<SiteReadViews> views;
    std::vector<std::string> errors;  // per site: why the device path could not take it ("" = fine)
    std::vector<std::vector<std::pair<common::p_Read, std::string>>> filtered;  // per site (BatchParameters::keep_filtered)
    struct Run;
    void runAll(BatchParameters const& prm);
    void finish(Run& run);
    std::unique_ptr<Run> pending;  // between submit() and collect()
    ~Impl();
};

SiteBatcher::SiteBatcher() : impl_(new Impl()) {}
SiteBatcher::~SiteBatcher() = default;  // (Impl::~Impl is defined behind Impl::Run)
size_t SiteBatcher::numSites() const { return impl_->graphs.size(); }
SiteCounts const& SiteBatcher::counts(size_t site) const { return impl_->counts.at(site); }

SiteReadViews const& SiteBatcher::views(size_t site) const { return impl_->views.at(site); }
std::string const& SiteBatcher::error(size_t site) const { return impl_->errors.at(site); }
std::vector<std::pair<common::p_Read, std::string>>& SiteBatcher::filtered(size_t site)
{
    if (impl_->filtered.size() < impl_->graphs.size())
        impl_->filtered.resize(impl_->graphs.size());
    return impl_->filtered.at(site);
}

size_t SiteBatcher::addSite(const Graph* graph, std::vector<common::p_Read>* reads, std::list<graphtools::Path> const* paths)
{
    impl_->graphs.push_back(graph);
    impl_->reads.push_back(reads);
    impl_->packed.push_back(nullptr);
    impl_->paths.push_back(paths);
    return impl_->graphs.size() - 1;
}

size_t SiteBatcher::addSite(const Graph* graph, PackedSite const* reads, std::list<graphtools::Path> const* paths)
{
    impl_->graphs.push_back(graph);
    impl_->reads.push_back(nullptr);
    impl_->packed.push_back(reads);
    impl_->paths.push_back(paths);
    return impl_->graphs.size() - 1;
}

// The state of one run(): the flat inputs handed to the device and the flat records that come back.  Its steps are the
// phases PG_BATCH_TIMING=1 reports.
struct SiteBatcher::Impl::Run
{
    Run(Impl& impl_, BatchParameters const& prm_) : impl(impl_), prm(prm_), n_sites(impl_.graphs.size()) {}
    ~Run()
    {
        // the batch object returns to the pool (with its device buffers when the run went through), the graph set goes
        if (batch)
            batchPool(prm.device).giveBack(ctx, batch, finished);
        if (G)
            pg_graphs_destroy(ctx, G);
    }
    void graphCsr();
    void packReads();
    void deviceSubmit();   // graphs + reads up, every stage queued: returns while the device works
    void deviceCollect();  // waits for the batch, fetches its records
    void resultsToReads();
    void viewsOfPackedSites();
    void siteTables();

    Impl& impl;
    const BatchParameters prm;  // (a copy: a submitted run outlives the call that made it)
    const size_t n_sites;
    pg_ctx* ctx = nullptr;
    pg_graphs* G = nullptr;
    pg_batch* batch = nullptr;
    bool finished = false;
    bool packed_mode = false;
    GraphCsr csr;
    // inputs of pg_batch_upload / pg_batch_set_fragments, all sites back to back
    // (page-locked staging from the process-wide pool: every copy of the device section is a DMA on the copy stream; the
    // lanes of a workflow hold one set each -- the double buffer of "pinned hipMemcpyAsync double-buffering")
    std::vector<uint64_t> site_read0, site_base0;
    pghost::PinnedVec<uint32_t> base_off, gor, frag;
    pghost::PinnedVec<uint8_t> is_rev;
    std::vector<Read*> flat;  // object sites only
    pghost::PinnedVec<char> bases;
    // what the device hands back
    pghost::PinnedVec<pg_result> res;
    pghost::PinnedVec<pg_op> ops;
    std::vector<uint64_t> seq_off;
    pghost::PinnedVec<uint32_t> table, path;
    pghost::PinnedVec<pg_read_support> sup;
    std::vector<uint64_t> label_ext;  // [read][csr.label_words - 1]: the label sets' words beyond pg_read_support.label_mask
    LabelSet labelSetOf(uint64_t i) const
    {
        LabelSet ls(sup[i].label_mask);
        for (uint32_t w = 1; w < csr.label_words; ++w)
            ls.w[w] = label_ext[i * (csr.label_words - 1) + (w - 1)];
        return ls;
    }
    pg_count_layout lay{};
};

void SiteBatcher::run(BatchParameters const& prm)
{
    const size_t n = impl_->graphs.size();
    impl_->counts.assign(n, SiteCounts());
    impl_->views.assign(n, SiteReadViews());
    impl_->errors.assign(n, std::string());
    impl_->filtered.clear();
    impl_->filtered.resize(n);
    if (n == 0)
        return;
    try
    {
        impl_->runAll(prm);
        return;
    }
    catch (OutsideEnvelope const& e)
    {
        // The reference has no bound on read length, nodes, columns or labels (gssw.c:527-786, GraphAligner.cpp:110-167,
        // ReadCounting.cpp:96-127); the device kernels do.  A site beyond them is reported, the run goes on.
        impl_->counts.assign(n, SiteCounts());
        impl_->views.assign(n, SiteReadViews());
        impl_->filtered.clear();
        impl_->filtered.resize(n);
        if (n == 1)
        {
            impl_->errors[0] = e.what();
            if (impl_->reads[0])
                impl_->reads[0]->clear();  // no read of this site is MAPPED (Align.cpp:155 keeps only those)
            return;
        }
    }
    // Some site of the batch is outside the envelope.  The batch is cut in two and each half run as a batch of its own (which
    // cuts itself again if it has to): the offending site ends up alone and reports, every other site comes out as if it were
    // not there -- after about 2 log2(n) smaller batches, not n one-site ones (a 128-site batch with one 65-label graph used to
    // become 128 uploads and device round trips, run one after the other).
    const size_t mid = n / 2;
    for (size_t part = 0; part < 2; ++part)
    {
        const size_t lo = part ? mid : 0, hi = part ? n : mid;
        SiteBatcher half;
        for (size_t s = lo; s < hi; ++s)
        {
            if (impl_->packed[s])
                half.addSite(impl_->graphs[s], impl_->packed[s], impl_->paths[s]);
            else
                half.addSite(impl_->graphs[s], impl_->reads[s], impl_->paths[s]);
        }
        half.run(prm);
        for (size_t s = lo; s < hi; ++s)
        {
            impl_->counts[s] = std::move(half.impl_->counts[s - lo]);
            impl_->views[s] = std::move(half.impl_->views[s - lo]);
            impl_->errors[s] = std::move(half.impl_->errors[s - lo]);
            if (s - lo < half.impl_->filtered.size())
                impl_->filtered[s] = std::move(half.impl_->filtered[s - lo]);
        }
    }
}

SiteBatcher::Impl::~Impl() = default;

namespace
{
// PG_BATCH_TIMING=1: wall-clock of the phases of a run on stderr
struct PhaseMarks
{
    const bool timing = std::getenv("PG_BATCH_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t_prev = std::chrono::steady_clock::now();
    void operator()(const char* what)
    {
        if (!timing)
            return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[SiteBatcher] %-18s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    }
};
}  // namespace

void SiteBatcher::Impl::runAll(BatchParameters const& prm)
{
    PhaseMarks mark;
    Impl::Run run(*this, prm);
    run.graphCsr();
    mark("graph csr");
    run.packReads();
    mark("pack reads");
    run.deviceSubmit();  // the only part under the device mutex
    mark("device submit");
    finish(run);
}

// what follows the queued stages: the records come down, the reads / views and the site tables are made
void SiteBatcher::Impl::finish(Run& run)
{
    PhaseMarks mark;
    run.deviceCollect();
    mark("device collect");
    if (run.packed_mode)
        run.viewsOfPackedSites();
    else
        run.resultsToReads();
    mark("results -> reads");
    run.siteTables();
    mark("site tables");
}

bool SiteBatcher::submit(BatchParameters const& prm)
{
    const size_t n = impl_->graphs.size();
    impl_->counts.assign(n, SiteCounts());
    impl_->views.assign(n, SiteReadViews());
    impl_->errors.assign(n, std::string());
    impl_->filtered.clear();
    impl_->filtered.resize(n);
    impl_->pending.reset();
    if (n == 0)
        return true;
    try
    {
        std::unique_ptr<Impl::Run> run(new Impl::Run(*impl_, prm));
        run->graphCsr();
        run->packReads();
        run->deviceSubmit();
        impl_->pending = std::move(run);
        return true;
    }
    catch (OutsideEnvelope const&)
    {
        return false;  // nothing of the attempt is left (the run released its device objects): run() isolates the site
    }
}

void SiteBatcher::collect()
{
    if (!impl_->pending)
        return;
    std::unique_ptr<Impl::Run> run = std::move(impl_->pending);
    impl_->finish(*run);
}

void SiteBatcher::Impl::Run::graphCsr()
{
    for (const Graph* g : impl.graphs)
        csr.add(*g);
    csr.finish();
}

void SiteBatcher::Impl::Run::packReads()
{
    // ---- pack the reads of all sites: offsets first, then every site fills its own slice --------------------
    packed_mode = impl.packed[0] != nullptr;
    site_read0.assign(n_sites + 1, 0);
    site_base0.assign(n_sites + 1, 0);
    for (size_t s = 0; s < n_sites; ++s)
        if ((impl.packed[s] != nullptr) != packed_mode || (!packed_mode && !impl.reads[s]))
            throw std::logic_error("SiteBatcher: a batch holds either object sites or packed sites");
    for (size_t s = 0; s < n_sites; ++s)
    {
        uint64_t site_bases = 0, site_reads = 0;
        if (packed_mode)
        {
            site_bases = impl.packed[s]->bases.size();
            site_reads = impl.packed[s]->size();
        }
        else
        {
            for (auto const& r : *impl.reads[s])
                site_bases += r->bases().size();
            site_reads = impl.reads[s]->size();
        }
        site_read0[s + 1] = site_read0[s] + site_reads;
        site_base0[s + 1] = site_base0[s] + site_bases;
    }
    if (site_read0[n_sites] > 0xFFFFFFFFull || site_base0[n_sites] > 0xFFFFFFFFull)
        throw std::runtime_error("SiteBatcher: more than 2^32 reads or bases in one batch");
    base_off.resize(site_read0[n_sites] + 1);
    base_off[0] = 0;
    gor.resize(site_read0[n_sites]);
    frag.resize(site_read0[n_sites]);
    is_rev.resize(site_read0[n_sites]);
    flat.resize(packed_mode ? 0 : site_read0[n_sites]);
    bases.resize(site_base0[n_sites]);
    pghost::parallelFor(
        n_sites, prm.threads,
        [&](size_t s) {
            uint64_t i = site_read0[s], at = site_base0[s];
            if (packed_mode)
            {
                PackedSite const& p = *impl.packed[s];
                std::copy(p.bases.begin(), p.bases.end(), bases.begin() + (std::ptrdiff_t)at);
                for (size_t k = 0; k < p.size(); ++k, ++i)
                {
                    base_off[i + 1] = (uint32_t)(at + p.base_end[k]);
                    gor[i] = (uint32_t)s;
                    frag[i] = p.fragment[k];
                    is_rev[i] = (p.flags[k] & PackedSite::REVERSE) ? 1 : 0;
                }
                return;
            }
            std::unordered_map<std::string, uint32_t> frag_ids;  // fragment ids are local to a site
            for (auto& r : *impl.reads[s])
            {
                flat[i] = r.get();
                std::copy(r->bases().begin(), r->bases().end(), bases.begin() + (std::ptrdiff_t)at);
                at += r->bases().size();
                base_off[i + 1] = (uint32_t)at;
                gor[i] = (uint32_t)s;
                frag[i] = frag_ids.emplace(r->fragment_id(), (uint32_t)frag_ids.size()).first->second;
                is_rev[i] = r->is_reverse_strand() ? 1 : 0;
                r->set_graph_mapping_status(Read::UNMAPPED);
                ++i;
            }
        },
        8);
}

void SiteBatcher::Impl::Run::deviceSubmit()
{
    // ---- device section: calls on one context are serialised; everything before and after overlaps across threads ------
    ctx = deviceContext(prm.device);
    const uint32_t n = (uint32_t)gor.size();
    seq_off.assign(n_sites + 1, 0);
    const bool timing = std::getenv("PG_BATCH_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!timing)
            return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[SiteBatcher]   %-16s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // Three steps, each of which another lane's batch can be in at the same time (paragraph_amd.h, "Threading"):
    //   1. graph set + indexes and then the reads go up the copy stream (work items and fragment tables are host work),
    //   2. the stage calls, which share the context's workspace (deviceMutex),
    //   3. the results come down; the batch object returns to the pool with its device buffers.
    check(ctx, pg_graphs_upload(ctx, (uint32_t)n_sites, csr.node_off.data(), csr.seq_off.data(), csr.seq.data(),
                                csr.pred_off.data(), csr.pred.empty() ? nullptr : csr.pred.data(), &G),
          "pg_graphs_upload");
    check(ctx, pg_graphs_set_labels_wide(ctx, G, csr.label_mask.empty() ? nullptr : csr.label_mask.data(), csr.label_words,
                                         csr.n_labels.data()),
          "pg_graphs_set_labels_wide");
    // ... and so are the per-graph indexes of the optional stages and of the KmerFilter
    std::vector<uint32_t> path_off{ 0 }, path_node_off{ 0 }, path_nodes;
    if ((prm.kmer_sequence_matching || prm.klib_sequence_matching) && n)
    {
        for (size_t s = 0; s < n_sites; ++s)
        {
            if (!impl.paths[s])
                throw std::logic_error("SiteBatcher: the k-mer and klib stages need the paths of every site");
            for (auto const& p : *impl.paths[s])
            {
                path_nodes.insert(path_nodes.end(), p.nodes.begin(), p.nodes.end());
                path_node_off.push_back((uint32_t)path_nodes.size());
            }
            path_off.push_back((uint32_t)path_node_off.size() - 1);
        }
        if (path_nodes.empty())
            path_nodes.push_back(0);
    }
    mark("graph set up");
    const bool shortcut = prm.exact_match_shortcut && !prm.path_sequence_matching && !prm.kmer_sequence_matching && !prm.klib_sequence_matching;
    if ((prm.path_sequence_matching || shortcut) && n)
        check(ctx, pg_graphs_build_path_index(ctx, G, 32), "pg_graphs_build_path_index");
    mark("path index");
    if (prm.kmer_sequence_matching && n)
        check(ctx, pg_graphs_build_kmer_index(ctx, G, 16, path_off.data(), path_node_off.data(), path_nodes.data()),
              "pg_graphs_build_kmer_index");
    if (prm.klib_sequence_matching && n)
        check(ctx, pg_graphs_build_klib_index(ctx, G, path_off.data(), path_node_off.data(), path_nodes.data()), "pg_graphs_build_klib_index");
    if (prm.kmer_len != 0)  // createReadFilter(graph, nonuniq, frac, kmer_len): NonUniq -> BadAlign -> KmerFilter (ReadFilter.cpp:74-90)
        check(ctx, pg_graphs_build_filter_index(ctx, G, prm.kmer_len, nullptr), "pg_graphs_build_filter_index");
    mark("graphs up");
    batch = batchPool(prm.device).take(ctx);
    check(ctx, pg_batch_upload(ctx, batch, G, n, gor.data(), base_off.data(), bases.data()), "pg_batch_upload");
    mark("batch upload");
    check(ctx, pg_batch_set_fragments(ctx, batch, frag.data(), is_rev.data()), "pg_batch_set_fragments");
    mark("reads up");
    std::unique_lock<std::mutex> lock(deviceMutex(prm.device));
    mark("wait for device");
    pg_count_params cp{};
    cp.remove_nonuniq = prm.remove_nonuniq_reads ? 1 : 0;
    cp.use_support_filters = prm.use_support_filters ? 1 : 0;
    cp.bad_align_frac = prm.bad_align_frac;
    if (prm.kmer_len != 0)
    {
        cp.use_kmer_filter = 1;  // the index was built above
    }
    uint32_t align_flags = prm.alignment_flags;
    // Seed stages of the cascade (CompositeAligner.cpp:78-150).  After each one the filter chain runs on the device (count
    // pass); a read that is MAPPED and accepted is done, everything else goes on to the next stage with the earlier
    // records kept.
    // The hand-over is decided on the device (pg_batch_retire_mapped): neither flags nor supports nor a mask cross to the host,
    // the work items of the stages behind it are re-written there, and the whole cascade is queued without a wait.
    uint32_t keep = 0;
    auto hand_over = [&]() {
        check(ctx, pg_batch_count(ctx, batch, &cp, nullptr), "pg_batch_count");
        check(ctx, pg_batch_retire_mapped(ctx, batch), "pg_batch_retire_mapped");
        keep = PG_AF_KEEP_RESULTS;
    };

    if (prm.path_sequence_matching && n)
    {
        check(ctx, pg_batch_path_align(ctx, batch), "pg_batch_path_align");
        hand_over();
    }
    else if (shortcut && n)
    {
        // not a stage of the cascade: no filter chain, no hand-over of MAPPED reads -- only the reads whose gssw record is
        // forced keep it (status without PG_STATUS_PATH_ALIGNER: the record of the gssw stage)
        check(ctx, pg_batch_path_align(ctx, batch), "pg_batch_path_align");
        check(ctx, pg_batch_retire_exact_matches(ctx, batch), "pg_batch_retire_exact_matches");
        keep = PG_AF_KEEP_RESULTS;
    }
    if (prm.kmer_sequence_matching && n)
    {
        check(ctx, pg_batch_kmer_align(ctx, batch, keep), "pg_batch_kmer_align");
        hand_over();
    }
    if (prm.klib_sequence_matching && n)
    {
        check(ctx, pg_batch_klib_align(ctx, batch, keep), "pg_batch_klib_align");
        hand_over();  // (the stage's overflow word is read when the batch is back: deviceCollect)
    }
    if (keep)  // (the extension flag is ignored when flags == PG_AF_ALL)
        align_flags = (align_flags & (PG_AF_CIGAR | PG_AF_BOTH_STRANDS | PG_AF_REVERSE_GRAPH)) | PG_AF_KEEP_RESULTS;
    check(ctx, pg_batch_align(ctx, batch, align_flags), "pg_batch_align");
    check(ctx, pg_batch_count(ctx, batch, &cp, nullptr), "pg_batch_count");
    lock.unlock();  // the kernels are queued; the next batch may queue its own behind them
    mark("align + count");
}

void SiteBatcher::Impl::Run::deviceCollect()
{
    const uint32_t n = (uint32_t)gor.size();
    const bool timing = std::getenv("PG_BATCH_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!timing)
            return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[SiteBatcher]   %-16s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };

    // one wait for the batch (its sizes are in page-locked memory by then), one for the five copies
    uint64_t n_ops = 0, n_path = 0;
    check(ctx, pg_graphs_count_layout(G, &lay), "pg_graphs_count_layout");
    check(ctx, pg_graphs_seq_offsets(G, seq_off.data()), "pg_graphs_seq_offsets");
    res.resize(n);
    sup.resize(n);
    table.resize(lay.n_counters);
    check(ctx, pg_batch_result_sizes(ctx, batch, &n_ops, &n_path), "pg_batch_result_sizes");
    mark("batch done");
    if (prm.klib_sequence_matching && n)
    {
        uint32_t overflow = 0;
        check(ctx, pg_graphs_klib_error(ctx, G, &overflow), "pg_graphs_klib_error");
        if (overflow)
            throw std::runtime_error("klib stage: CIGAR buffer overflow on the device");
    }
    ops.resize(n_ops + 1);
    path.resize(n_path + 1);
    check(ctx, pg_batch_download_all(ctx, batch, res.data(), ops.data(), ops.size(), table.data(), sup.data(), path.data(), path.size()),
          "pg_batch_download_all");
    if (csr.label_words > 1)
    {
        label_ext.assign((size_t)n * (csr.label_words - 1), 0);
        check(ctx, pg_batch_download_label_ext(ctx, batch, label_ext.data(), label_ext.size()), "pg_batch_download_label_ext");
    }
    finished = true;
    mark("results down");
}

void SiteBatcher::Impl::Run::resultsToReads()
{
    // --validate-alignments: per site the node lists of its paths, keyed by path id, as ValidationAligner's constructor builds them
    std::vector<std::unordered_map<std::string, std::string>> validation_paths;
    if (prm.validate_alignments)
    {
        validation_paths.resize(n_sites);
        for (size_t s = 0; s < n_sites; ++s)
        {
            if (!impl.paths[s])
                throw std::logic_error("SiteBatcher: validate_alignments needs the paths of every site");
            for (auto const& p : *impl.paths[s])
            {
                std::string& nodes = validation_paths[s][p.encode()];
                nodes.clear();
                for (auto nd : p.nodeIds())
                    nodes += (nodes.empty() ? "" : "->") + std::to_string(nd);
            }
        }
    }
    // ---- fan results back into the reads ---------------------------------------------------------------
    const uint32_t n = (uint32_t)gor.size();
    pghost::parallelFor(
        n, prm.threads,
        [&](size_t i) {
            Read& read = *flat[i];
            if (prm.validate_alignments && !read.bases().empty() && sup[i].status == 0)
                validationAccount(read, validation_paths[gor[i]]);  // went through the aligner and stayed unmapped: counts in the total
            if (read.bases().empty() || sup[i].status == 0)
                return;
            if (sup[i].status == 3)
                throw std::runtime_error("invalid alignment on the device path for fragment " + read.fragment_id());
            if (res[i].status & PG_STATUS_PATH_ALIGNER)
            {
                // PathAligner.cpp:121-161: the match's own strand, bases replaced, qualities untouched
                if (res[i].returned_reverse)
                    read.set_bases(reverseComplement(read.bases()));
                read.set_is_graph_reverse_strand(res[i].returned_reverse != 0);
                std::string buf(16 + 12 * (size_t)res[i].n_ops, '\0');
                buf.resize(pg_render_cigar(&res[i], ops.data(), &buf[0], buf.size()));
                read.set_graph_cigar(buf);
                read.set_graph_pos(res[i].graph_pos);
                read.set_graph_alignment_score(res[i].score);
                read.set_is_graph_alignment_unique(res[i].is_unique != 0);
                read.set_graph_mapq(res[i].mapq);
            }
            else  // the k-mer and klib stages replace the bases of a reverse hit but leave the qualities as they are
                applyResult(read, res[i], ops.data(), true, !(res[i].status & (PG_STATUS_KMER_ALIGNER | PG_STATUS_KLIB_ALIGNER)));
            read.set_graph_mapping_status(sup[i].status == 1 ? Read::MAPPED : Read::BAD_ALIGN);
            if (prm.validate_alignments)
                validationAccount(read, validation_paths[gor[i]]);
            read.clear_graph_nodes_supported();
            read.clear_graph_edges_supported();
            read.clear_graph_sequences_supported();
            if (sup[i].status != 1)
                return;
            const Graph& g = *impl.graphs[gor[i]];
            uint32_t prev = 0;
            std::vector<std::pair<std::string, std::string>> edges;
            for (uint32_t k = 0; k < sup[i].n_path; ++k)
            {
                const uint32_t en = path[sup[i].path_off + k];
                const uint32_t nd = PG_PATH_NODE(en);
                if (PG_PATH_NODE_OK(en))
                    read.add_graph_nodes_supported(g.nodeName(nd));
                if (k > 0 && PG_PATH_EDGE_OK(en))
                    edges.emplace_back(g.nodeName(prev), g.nodeName(nd));
                prev = nd;
            }
            std::sort(edges.begin(), edges.end());  // the reference collects them in a std::set of name pairs
            for (auto const& e : edges)
                read.add_graph_edges_supported(e.first + "_" + e.second);
            const auto& names = csr.label_names[gor[i]];
            const LabelSet labels = labelSetOf(i);
            for (size_t b = 0; b < names.size(); ++b)
                if (labels.test(b))
                    read.add_graph_sequences_supported(names[b]);
        },
        512);
}

void SiteBatcher::Impl::Run::viewsOfPackedSites()
{
    // ---- packed sites: what the statistics need of the MAPPED reads, straight from the device records ----------
    pghost::parallelFor(
        n_sites, prm.threads,
        [&](size_t s) {
            PackedSite const& p = *impl.packed[s];
            SiteReadViews& v = impl.views[s];
            v.label_names = csr.label_names[s];
            v.reads.reserve(p.size());  // (most reads of a site map; a read crosses two or three nodes)
            v.pieces.reserve(2 * p.size() + 8);
            v.support.reserve(3 * p.size() + 8);
            uint32_t n_fragments = 0;
            for (size_t k = 0; k < p.size(); ++k)
            {
                const size_t i = site_read0[s] + k;
                n_fragments = std::max(n_fragments, p.fragment[k] + 1);
                if (p.readLength(k) == 0 || sup[i].status == 0)
                    continue;
                if (sup[i].status == 3)
                    throw std::runtime_error("invalid alignment on the device path in site " + std::to_string(s));
                if (sup[i].status != 1)
                    continue;
                MappedReadView m;
                m.fragment = p.fragment[k];
                m.read_length = p.readLength(k);
                m.chrom_id = p.chrom_id[k];
                m.pos = p.pos[k];
                m.mate_chrom_id = p.mate_chrom_id[k];
                m.mate_pos = p.mate_pos[k];
                m.is_mapped = (p.flags[k] & PackedSite::MAPPED) != 0;
                m.is_mate_mapped = (p.flags[k] & PackedSite::MATE_MAPPED) != 0;
                m.is_reverse_strand = (p.flags[k] & PackedSite::REVERSE) != 0;
                m.is_mate_reverse_strand = (p.flags[k] & PackedSite::MATE_REVERSE) != 0;
                m.is_graph_mapped = true;
                m.is_graph_reverse_strand = (res[i].status & PG_STATUS_PATH_ALIGNER) ? res[i].returned_reverse != 0
                                                                                     : m.is_reverse_strand != (res[i].returned_reverse != 0);
                m.graph_pos = res[i].graph_pos;
                m.graph_alignment_score = res[i].score;
                m.pieces_off = (uint32_t)v.pieces.size();
                for (uint32_t o = 0; o < res[i].n_ops; ++o)
                {
                    const pg_op op = ops[res[i].ops_off + o];
                    const NodeId node = PG_OP_NODE(op);
                    if (v.pieces.size() == m.pieces_off || v.pieces.back().node != node)
                    {
                        v.pieces.emplace_back();
                        v.pieces.back().node = node;
                    }
                    NodeAlignment& na = v.pieces.back();
                    const uint32_t len = PG_OP_LEN(op);
                    switch (PG_OP_CODE(op))
                    {
                    case PG_OPC_M: na.matched += len; break;
                    case PG_OPC_X: na.mismatched += len; break;
                    case PG_OPC_N: na.missing += len; break;
                    case PG_OPC_I: na.inserted += len; break;
                    case PG_OPC_D: na.deleted += len; break;
                    case PG_OPC_S: na.clipped += len; break;
                    default: break;  // PG_OPC_EMPTY: the node is on the path with an empty CIGAR
                    }
                }
                m.n_pieces = (uint32_t)v.pieces.size() - m.pieces_off;
                m.sequences = labelSetOf(i);
                m.support_off = (uint32_t)v.support.size();
                m.n_support = sup[i].n_path;
                v.support.insert(v.support.end(), path.begin() + sup[i].path_off, path.begin() + sup[i].path_off + sup[i].n_path);
                v.reads.push_back(m);
            }
            v.n_fragments = n_fragments;
        },
        8);
}

void SiteBatcher::Impl::Run::siteTables()
{
    // ---- per-site tables --------------------------------------------------------------------------------
    auto entry = [&](uint64_t off) {
        CountEntry e;
        e.count = table[off];
        e.reads = table[off + 1];
        e.fwd = table[off + 2];
        e.rev = table[off + 3];
        return e;
    };
    pghost::parallelFor(
        n_sites, prm.threads,
        [&](size_t s) {
            const Graph& g = *impl.graphs[s];
            SiteCounts& sc = impl.counts[s];
            const uint32_t nb = csr.node_off[s];
            for (NodeId nd = 0; nd != g.numNodes(); ++nd)
            {
                if (prm.node_counts)
                {
                    CountEntry e = entry(lay.node_base + 4ull * (nb + nd));
                    if (e.count)
                        sc.by_node[g.nodeName(nd)] = e;
                }
                for (uint32_t q = csr.pred_off[nb + nd]; q < csr.pred_off[nb + nd + 1]; ++q)
                {
                    CountEntry ee = entry(lay.edge_base + 4ull * q);
                    if (ee.count)
                        sc.by_edge[g.nodeName(csr.pred[q]) + "_" + g.nodeName(nd)] = ee;
                }
            }
            const auto& names = csr.label_names[s];
            for (uint64_t m = 1; prm.sequence_counts && m < seq_off[s + 1] - seq_off[s]; ++m)
            {
                CountEntry e = entry(lay.seq_base + 4ull * (seq_off[s] + m));
                if (!e.count)
                    continue;
                std::string key;
                for (size_t b = 0; b < names.size(); ++b)
                    if ((m >> b) & 1)
                        key += (key.empty() ? "" : ",") + names[b];  // names are sorted -> sorted join
                sc.by_sequence[key] = e;
            }
            const uint64_t t = lay.tally_base + 4ull * s;
            sc.aligned = table[t] & 0x7FFFFFFFu;
            sc.mapped = table[t + 1];
            sc.bad_align = table[t + 2];
            sc.nonuniq = table[t + 3];
            if (prm.sequence_counts && names.size() > PG_MAX_SEQ_TABLE_LABELS)
            {
                // the device keeps a dense sequence-set table only for graphs with <= 8 labels (2^labels slots); for the others the
                // totals of countPathFamilies (ReadCounting.cpp:96-127) are summed here from the per-read label sets: a fragment
                // supports the union of its MAPPED reads' sets
                struct Agg
                {
                    LabelSet labels;
                    uint32_t n = 0, fwd = 0, rev = 0;
                };
                std::map<uint32_t, Agg> fragments;
                for (uint64_t i = site_read0[s]; i < site_read0[s + 1]; ++i)
                {
                    if (sup[i].status != 1)
                        continue;
                    Agg& f = fragments[frag[i]];
                    f.labels |= labelSetOf(i);
                    ++f.n;
                    const bool graph_rev = (res[i].status & PG_STATUS_PATH_ALIGNER) ? res[i].returned_reverse != 0
                                                                                    : (is_rev[i] != 0) != (res[i].returned_reverse != 0);
                    ++(graph_rev ? f.rev : f.fwd);
                }
                for (auto const& kv : fragments)
                {
                    if (!kv.second.labels.any())
                        continue;
                    std::string key;
                    for (size_t b = 0; b < names.size(); ++b)
                        if (kv.second.labels.test(b))
                            key += (key.empty() ? "" : ",") + names[b];
                    CountEntry& e = sc.by_sequence[key];
                    e.count += 1;
                    e.reads += kv.second.n;
                    e.fwd += kv.second.fwd;
                    e.rev += kv.second.rev;
                }
            }
            if (packed_mode)
                return;
            // only MAPPED reads survive (Align.cpp:155); the ones a filter rejected go to filtered() when asked for
            // (Disambiguation.cpp:183-203: status BAD_ALIGN, the filter's message under "error")
            static const char* const kFilterName[] = { "", "nonuniq", "bad_align", "kmer_tooshort", "kmer_uncov" };
            std::vector<common::p_Read> kept;
            uint64_t i = site_read0[s];
            for (auto& r : *impl.reads[s])
            {
                const uint64_t at = i++;
                if (!r->bases().empty() && r->graph_mapping_status() == Read::MAPPED)
                    kept.emplace_back(std::move(r));
                else if (prm.keep_filtered && !r->bases().empty() && sup[at].status == 2)
                    impl.filtered[s].emplace_back(std::move(r), kFilterName[sup[at].filter <= 4 ? sup[at].filter : 0]);
            }
            impl.reads[s]->swap(kept);
        },
        8);
}
}  // namespace paragraph
