// Internal: flat arrays in page-locked host memory for the hand-over to the device (north_star: "pinned hipMemcpyAsync
// double-buffering").  The reference fills std::vector<p_Read> per site on worker threads
// (src/c++/lib/grmpy/AlignSamples.cpp:115-172, src/c++/lib/common/ReadExtraction.cpp:38-219); here the reads of a whole
// batch are packed into flat arrays, and when those are pinned every copy of pg_batch_upload / pg_batch_download* is a DMA
// on the copy stream beside another lane's kernels.  Blocks come from a process-wide pool (hipHostMalloc costs
// milliseconds per block, and a workflow needs the same sizes batch after batch); two lanes in flight = two sets of
// staging buffers = the double buffer.  If page-locking fails (RLIMIT_MEMLOCK) the block is ordinary memory: slower
// copies, same results.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace pghost
{
struct PinnedBlock
{
    void* p = nullptr;
    size_t bytes = 0;
    bool pinned = false;
};
PinnedBlock pinnedTake(size_t bytes);    // at least `bytes`, contents undefined
void pinnedGive(PinnedBlock const& blk); // back to the pool (or freed when the pool is full)
size_t pinnedBytesAllocated();            // page-locked bytes obtained so far (tests)

template <typename T> class PinnedVec
{
public:
    PinnedVec() = default;
    PinnedVec(PinnedVec const&) = delete;
    PinnedVec& operator=(PinnedVec const&) = delete;
    ~PinnedVec() { release(); }
    // contents are NOT preserved and NOT initialised
    void resize(size_t n)
    {
        if (n * sizeof(T) > blk_.bytes)
        {
            release();
            blk_ = pinnedTake(n * sizeof(T));
        }
        n_ = n;
    }
    void assign(size_t n, T const& v)
    {
        resize(n);
        if (sizeof(T) == 1)
            memset(blk_.p, (int)(unsigned char)*(const unsigned char*)&v, n);
        else
            for (size_t i = 0; i < n; ++i)
                data()[i] = v;
    }
    void release()
    {
        if (blk_.p)
            pinnedGive(blk_);
        blk_ = PinnedBlock();
        n_ = 0;
    }
    T* data() { return (T*)blk_.p; }
    T const* data() const { return (T const*)blk_.p; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    bool pinned() const { return blk_.pinned; }
    T& operator[](size_t i) { return data()[i]; }
    T const& operator[](size_t i) const { return data()[i]; }
    T* begin() { return data(); }
    T* end() { return data() + n_; }
    T const* begin() const { return data(); }
    T const* end() const { return data() + n_; }

private:
    PinnedBlock blk_;
    size_t n_ = 0;
};
}  // namespace pghost
