// Sample-side input of the host workflow: coordinates, FASTA, BGZF/BAM/BAI, read-pair bookkeeping and per-site read
// extraction.  Headers under include/common cite the reference interfaces each class mirrors.
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>
#include <immintrin.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "common/BamReader.hh"
#include "common/Fasta.hh"
#include "common/ReadExtraction.hh"
#include "common/ReadPairs.hh"
#include "common/Region.hh"
#include "paragraph/PackedReads.hh"

#include "inflate.hh"

namespace common
{
// ------------------------------------------------------------------------------------------------------------------
// coordinates
// ------------------------------------------------------------------------------------------------------------------
void parsePos(std::string const& text, std::string& chrom, int64_t& start, int64_t& end)
{
    // fields are separated by any of " :-"; empty fields are dropped (stringutil::split)
    std::vector<std::string> fields;
    std::string cur;
    for (char c : text)
    {
        if (c == ' ' || c == ':' || c == '-')
        {
            if (!cur.empty())
                fields.push_back(cur);
            cur.clear();
        }
        else
            cur += c;
    }
    if (!cur.empty())
        fields.push_back(cur);
    auto number = [](std::string s) {
        s.erase(std::remove(s.begin(), s.end(), ','), s.end());
        return (int64_t)std::stoll(s);
    };
    if (fields.size() >= 1)
        chrom = fields[0];
    if (fields.size() >= 2)
        start = number(fields[1]) - 1;
    if (fields.size() >= 3)
        end = number(fields[2]) - 1;
}

std::string formatPos(std::string const& chrom, int64_t start, int64_t end)
{
    std::string out = chrom;
    if (start >= 0)
    {
        out += ":" + std::to_string(start + 1);
        if (end >= 0)
            out += "-" + std::to_string(end + 1);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// FASTA
// ------------------------------------------------------------------------------------------------------------------
struct FastaFile::Impl
{
    struct Contig
    {
        size_t length, offset, line_bases, line_bytes;
    };
    std::string filename;
    int fd = -1;  // queries use pread: one FastaFile can serve many threads
    std::unordered_map<std::string, Contig> contigs;
    std::vector<std::string> order;
    ~Impl()
    {
        if (fd >= 0)
            ::close(fd);
    }
};

FastaFile::FastaFile(std::string const& path) : impl_(new Impl)
{
    impl_->filename = path;
    impl_->fd = ::open(path.c_str(), O_RDONLY);
    if (impl_->fd < 0)
        throw std::runtime_error("Cannot open FASTA file " + path);
    std::ifstream fai(path + ".fai");
    if (!fai.good())
    {
        // no samtools index next to the file: derive one by scanning (the original indexes on the fly as well)
        std::string line, name;
        Impl::Contig c{};
        bool open = false;
        size_t at = 0;
        auto close = [&] {
            if (open)
            {
                impl_->contigs[name] = c;
                impl_->order.push_back(name);
            }
        };
        std::ifstream scan(path, std::ios::binary);
        while (std::getline(scan, line))
        {
            const size_t bytes = line.size() + 1;
            if (!line.empty() && line.back() == '\r')
                line.pop_back();
            if (!line.empty() && line[0] == '>')
            {
                close();
                name = line.substr(1, line.find_first_of(" \t") == std::string::npos ? std::string::npos : line.find_first_of(" \t") - 1);
                c = Impl::Contig{ 0, at + bytes, 0, 0 };
                open = true;
            }
            else if (open)
            {
                if (c.line_bases == 0 && !line.empty())
                {
                    c.line_bases = line.size();
                    c.line_bytes = bytes;
                }
                c.length += line.size();
            }
            at += bytes;
        }
        close();
        for (auto& kv : impl_->contigs)
            if (kv.second.line_bases == 0)
                kv.second.line_bases = kv.second.line_bytes = 1;
        return;
    }
    std::string line;
    while (std::getline(fai, line))
    {
        if (line.empty())
            continue;
        std::stringstream ss(line);
        std::string name;
        Impl::Contig c{};
        std::getline(ss, name, '\t');
        ss >> c.length >> c.offset >> c.line_bases >> c.line_bytes;
        if (ss.fail() || c.line_bases == 0 || c.line_bytes < c.line_bases)
            throw std::runtime_error("Malformed FASTA index line: " + line);
        impl_->contigs[name] = c;
        impl_->order.push_back(name);
    }
}

FastaFile::~FastaFile() = default;
std::string const& FastaFile::getFilename() const { return impl_->filename; }
std::vector<std::string> FastaFile::getContigNames() const { return impl_->order; }

size_t FastaFile::contigSize(std::string const& contig) const
{
    if (contig.empty())
    {
        size_t all = 0;
        for (auto const& kv : impl_->contigs)
            all += kv.second.length;
        return all;
    }
    auto it = impl_->contigs.find(contig);
    if (it == impl_->contigs.end())
        throw std::runtime_error("Contig " + contig + " is not known");
    return it->second.length;
}

std::string FastaFile::query(std::string const& location) const
{
    std::string chrom;
    int64_t start = -1, end = -1;
    parsePos(location, chrom, start, end);
    return query(chrom, start, end);
}

std::string FastaFile::query(std::string const& chrom, int64_t start, int64_t end) const
{
    if (end < start)
        return "";
    start = std::max<int64_t>(start, 0);
    auto it = impl_->contigs.find(chrom);
    if (it == impl_->contigs.end())
        throw std::runtime_error("Contig " + chrom + " is not known");
    Impl::Contig const& c = it->second;
    if ((size_t)start >= c.length)
        return "";
    const size_t last = std::min<size_t>((size_t)end, c.length - 1);
    // bytes [first_byte, last_byte] of the file cover the bases plus the line ends between them
    const size_t first_byte = c.offset + ((size_t)start / c.line_bases) * c.line_bytes + (size_t)start % c.line_bases;
    const size_t last_byte = c.offset + (last / c.line_bases) * c.line_bytes + last % c.line_bases;
    std::string raw(last_byte - first_byte + 1, '\0');
    for (size_t got = 0; got < raw.size();)
    {
        const ssize_t n = ::pread(impl_->fd, &raw[got], raw.size() - got, (off_t)(first_byte + got));
        if (n <= 0)
            throw std::runtime_error("Short read from FASTA file " + impl_->filename);
        got += (size_t)n;
    }
    std::string out;
    out.reserve(last - (size_t)start + 1);
    for (char ch : raw)
    {
        if (ch == '\n' || ch == '\r')
            continue;
        char u = (char)toupper((unsigned char)ch);
        out += (u == 'A' || u == 'C' || u == 'G' || u == 'T') ? u : 'N';
    }
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// BGZF: a series of gzip members of <= 64 KiB each; a virtual offset is (file offset of the member << 16) | offset
// inside its inflated payload (SAM spec 4.1)
// ------------------------------------------------------------------------------------------------------------------
namespace
{
// CRC-32 of a BGZF block.  zlib 1.2.11's crc32 runs at ~1 GB/s, about as long as inflating the block took; folding with
// carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel
// 2009: 64 bytes per step, then 128 -> 64 -> 32 bits with a Barrett reduction; constants for the reflected polynomial
// 0xEDB88320) does 20 GB/s.  Checked against zlib on 20 000 random buffers of every length class; without the instruction
// (or for the tail below 16 bytes) zlib's own routine is used.
__attribute__((target("pclmul,sse4.1")))
static uint32_t crc32Fold(const unsigned char* buf, size_t len, uint32_t crc)
{  // len >= 64 and a multiple of 16; crc in the register convention (already inverted)
    static const uint64_t __attribute__((aligned(16))) k1k2[] = { 0x0154442bd4ull, 0x01c6e41596ull };
    static const uint64_t __attribute__((aligned(16))) k3k4[] = { 0x01751997d0ull, 0x00ccaa009eull };
    static const uint64_t __attribute__((aligned(16))) k5k0[] = { 0x0163cd6124ull, 0 };
    static const uint64_t __attribute__((aligned(16))) poly[] = { 0x01db710641ull, 0x01f7011641ull };
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 16));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 32));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64; len -= 64;
    while (len >= 64)
    {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128((const __m128i*)(buf + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128((const __m128i*)(buf + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128((const __m128i*)(buf + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128((const __m128i*)(buf + 48)));
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i*)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16)
    {
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i*)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
static uint32_t crc32Fast(const unsigned char* p, size_t n)
{
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    if (n >= 64 && __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"))
    {
        const size_t body = n & ~(size_t)15;
        crc = ~crc32Fold(p, body, ~crc);
        p += body; n -= body;
    }
    return n ? (uint32_t)crc32(crc, p, (uInt)n) : crc;
}

// BAM's 4-bit base codes -> characters, 32 bases per step where the CPU has byte shuffles (two table look-ups by pshufb and an
// interleave), two per step otherwise
__attribute__((target("ssse3"))) static void unpackBasesSsse3(const unsigned char* packed, uint32_t n, char* out)
{
    const __m128i table = _mm_setr_epi8('=', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N');
    const __m128i low = _mm_set1_epi8(0x0F);
    uint32_t i = 0;
    for (; i + 32 <= n; i += 32)
    {
        const __m128i v = _mm_loadu_si128((const __m128i*)(packed + i / 2));
        const __m128i hi = _mm_shuffle_epi8(table, _mm_and_si128(_mm_srli_epi16(v, 4), low));
        const __m128i lo = _mm_shuffle_epi8(table, _mm_and_si128(v, low));
        _mm_storeu_si128((__m128i*)(out + i), _mm_unpacklo_epi8(hi, lo));
        _mm_storeu_si128((__m128i*)(out + i + 16), _mm_unpackhi_epi8(hi, lo));
    }
    static const char kBases[] = "=ACMGRSVTWYHKDBN";
    for (; i < n; ++i)
        out[i] = kBases[(i & 1) ? packed[i / 2] & 0xF : packed[i / 2] >> 4];
}

static void unpackBases(const unsigned char* packed, uint32_t n, char* out)
{
    static const bool ssse3 = __builtin_cpu_supports("ssse3");
    if (ssse3)
    {
        unpackBasesSsse3(packed, n, out);
        return;
    }
    static const char kBases[] = "=ACMGRSVTWYHKDBN";
    for (uint32_t i = 0; i < n; ++i)
        out[i] = kBases[(i & 1) ? packed[i / 2] & 0xF : packed[i / 2] >> 4];
}

// a byte buffer that is never zero-filled and keeps slack behind its logical size (the block decoder stores whole words)
struct BlockBuffer
{
    std::unique_ptr<unsigned char[]> p;
    size_t cap = 0, n = 0;
    unsigned char* data() { return p.get(); }
    const unsigned char* data() const { return p.get(); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
    void resize(size_t want, size_t slack)
    {
        if (want + slack > cap)
        {
            cap = std::max<size_t>(want + slack, 65536 + 64);
            p.reset(new unsigned char[cap]);
        }
        n = want;
    }
    void swap(BlockBuffer& o)
    {
        p.swap(o.p);
        std::swap(cap, o.cap);
        std::swap(n, o.n);
    }
};

class Bgzf
{
public:
    explicit Bgzf(std::string const& path) : path_(path)
    {
        fp_ = fopen(path.c_str(), "rb");
        if (!fp_)
            throw std::runtime_error("ERROR: Failed to open " + path);
    }
    ~Bgzf()
    {
        if (zs_ready_)
            inflateEnd(&zs_);
        if (fp_)
            fclose(fp_);
    }
    Bgzf(Bgzf const&) = delete;
    Bgzf& operator=(Bgzf const&) = delete;

    void seek(uint64_t voffset)
    {
        const uint64_t coff = voffset >> 16;
        if (!(have_block_ && coff == block_start_))
            loadBlock(coff);
        within_ = (size_t)(voffset & 0xFFFF);
        if (within_ > data_.size())
            throw std::runtime_error("BGZF: virtual offset beyond block in " + path_);
    }
    uint64_t tell()
    {
        // a position at the very end of a block is the start of the next one (as bgzf_tell reports after a read); the next
        // block is NOT loaded for that -- pointers handed out by take() stay on the current block
        if (have_block_ && within_ == data_.size() && !eof_)
            return (block_start_ + block_csize_) << 16;
        return (block_start_ << 16) | (uint64_t)within_;
    }
    // n contiguous bytes of the current block without a copy (nullptr when they are not all in it: read() then); the pointer
    // is good until the block after the next one is loaded
    const unsigned char* take(size_t n)
    {
        if (!have_block_ || eof_ || within_ + n > data_.size())
            return nullptr;
        const unsigned char* p = data_.data() + within_;
        within_ += n;
        return p;
    }
    // returns the number of bytes read (< n only at end of file)
    size_t read(void* dst, size_t n)
    {
        size_t done = 0;
        while (done < n)
        {
            if (!have_block_)
                loadBlock(0);
            if (within_ == data_.size())
            {
                if (eof_)
                    break;
                loadBlock(block_start_ + block_csize_);
                within_ = 0;
                continue;
            }
            const size_t take = std::min(n - done, data_.size() - within_);
            memcpy((char*)dst + done, data_.data() + within_, take);
            within_ += take;
            done += take;
        }
        return done;
    }
    void readExact(void* dst, size_t n, const char* what)
    {
        if (read(dst, n) != n)
            throw std::runtime_error(std::string("Truncated BAM (") + what + ") in " + path_);
    }

private:
    // Inflated blocks are kept in a small ring: neighbouring region queries (sites a few kbp apart, a mate lookup in
    // the middle of a scan) come back to the same 64 KiB blocks again and again.
    void loadBlock(uint64_t coff)
    {
        within_ = 0;
        const bool have_current = have_block_ && !eof_ && !data_.empty();
        have_block_ = true;
        for (Cached& c : ring_)
        {
            if (!(c.valid && c.start == coff))
                continue;
            // trade places: the block asked for becomes current, the current one takes its slot
            std::swap(block_start_, c.start);
            std::swap(block_csize_, c.csize);
            data_.swap(c.data);
            c.valid = have_current;
            eof_ = false;
            return;
        }
        if (have_current)
            stash();
        inflateBlock(coff);
    }

    // park the current block in the ring before its buffer is reused
    void stash()
    {
        Cached& slot = ring_[ring_next_];
        ring_next_ = (ring_next_ + 1) % kRing;
        slot.valid = true;
        slot.start = block_start_;
        slot.csize = block_csize_;
        slot.data.swap(data_);
    }

    void inflateBlock(uint64_t coff)
    {
        block_start_ = coff;
        block_csize_ = 0;
        data_.clear();
        if (fseeko(fp_, (off_t)coff, SEEK_SET) != 0)
            throw std::runtime_error("BGZF: seek failed in " + path_);
        unsigned char hdr[12];
        const size_t got = fread(hdr, 1, sizeof hdr, fp_);
        if (got == 0)
        {
            eof_ = true;
            return;
        }
        if (got != sizeof hdr || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4))
            throw std::runtime_error("BGZF: bad block header in " + path_);
        const unsigned xlen = hdr[10] | (hdr[11] << 8);
        extra_.resize(xlen);
        if (fread(extra_.data(), 1, xlen, fp_) != xlen)
            throw std::runtime_error("BGZF: truncated block header in " + path_);
        int bsize = -1;
        for (size_t i = 0; i + 4 <= xlen;)
        {
            const unsigned slen = extra_[i + 2] | (extra_[i + 3] << 8);
            if (extra_[i] == 'B' && extra_[i + 1] == 'C' && slen == 2 && i + 6 <= xlen)
                bsize = extra_[i + 4] | (extra_[i + 5] << 8);
            i += 4 + slen;
        }
        if (bsize < 0)
            throw std::runtime_error("BGZF: block without BC field in " + path_);
        block_csize_ = (uint64_t)bsize + 1;
        if (block_csize_ < 12 + (uint64_t)xlen + 8)
            throw std::runtime_error("BGZF: bad block size in " + path_);
        const size_t cdata_len = block_csize_ - 12 - xlen - 8;
        cdata_.resize(cdata_len + 8, 0);  // the 8 trailer bytes (CRC32, ISIZE) double as the decoder's read slack
        if (fread(cdata_.data(), 1, cdata_len + 8, fp_) != cdata_len + 8)
            throw std::runtime_error("BGZF: truncated block in " + path_);
        const unsigned char* tail = cdata_.data() + cdata_len;
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        data_.resize(isize, 16);
        if (isize)
        {
            // own block decoder (inflate.hh); PG_BGZF_ZLIB=1 switches back to zlib's inflate for A/B timing
            static const bool use_zlib = std::getenv("PG_BGZF_ZLIB") != nullptr;
            if (!use_zlib)
            {
                if (pginflate::inflateBlock(cdata_.data(), cdata_len, data_.data(), isize) != pginflate::kOk)
                    throw std::runtime_error("BGZF: inflate failed in " + path_);
            }
            else
            {
                if (!zs_ready_)
                {
                    memset(&zs_, 0, sizeof zs_);
                    if (inflateInit2(&zs_, -15) != Z_OK)
                        throw std::runtime_error("BGZF: inflateInit2 failed");
                    zs_ready_ = true;
                }
                else
                    inflateReset(&zs_);
                zs_.next_in = cdata_.data();
                zs_.avail_in = (uInt)cdata_len;
                zs_.next_out = data_.data();
                zs_.avail_out = (uInt)isize;
                const int rc = inflate(&zs_, Z_FINISH);
                if (rc != Z_STREAM_END || zs_.avail_out != 0)
                    throw std::runtime_error("BGZF: inflate failed in " + path_);
            }
            const uint32_t want_crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
            if (crc32Fast(data_.data(), isize) != want_crc)
                throw std::runtime_error("BGZF: CRC mismatch in " + path_);
        }
        eof_ = false;
    }

    enum { kRing = 8 };
    struct Cached
    {
        bool valid = false;
        uint64_t start = 0, csize = 0;
        BlockBuffer data;
    };
    Cached ring_[kRing];
    size_t ring_next_ = 0;
    std::vector<unsigned char> extra_, cdata_;
    z_stream zs_;
    bool zs_ready_ = false;
    std::string path_;
    FILE* fp_ = nullptr;
    bool have_block_ = false, eof_ = false;
    uint64_t block_start_ = 0, block_csize_ = 0;
    BlockBuffer data_;
    size_t within_ = 0;
};

inline uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t le64(const unsigned char* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

struct Chunk
{
    uint64_t beg, end;
};

struct RefIndex
{
    std::unordered_map<uint32_t, std::vector<Chunk>> bins;
    std::vector<uint64_t> linear;  // smallest virtual offset of a record overlapping each 16 kbp window
};

// the UCSC binning scheme of the BAI (SAM spec 5.3): all bins a query [beg, end) can have records in
void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>& bins)
{
    --end;
    bins.push_back(0);
    for (int level = 1, shift = 26, first = 1; level <= 5; ++level, shift -= 3)
    {
        for (int64_t k = first + (beg >> shift); k <= first + (end >> shift); ++k)
            bins.push_back((uint32_t)k);
        first += 1 << (3 * level);
    }
}

struct BamRecord
{
    int32_t tid = -1, pos = -1, mtid = -1, mpos = -1;
    uint16_t flag = 0;
    uint8_t mapq = 0;
    int64_t end_pos = 0;  // pos + reference span of the CIGAR, at least pos + 1 (bam_endpos)
    // where the text fields sit in the reader's record buffer; decoded only for records that are handed out
    uint32_t name_at = 0, name_len = 0, seq_at = 0, seq_len = 0;
    std::string name, bases, quals;
};

struct RegionCursor
{
    bool valid = false;
    int32_t tid = -1;
    int64_t beg = 0, end = 0;
    std::vector<Chunk> chunks;
    size_t chunk = 0;
    uint64_t at = 0;  // virtual offset of the next record to look at
    bool started = false, finished = true;
};
}  // namespace

// header and index of one BAM: parsed once per process and shared by every reader of that file (the workflow opens one
// reader per worker thread)
struct BamMeta
{
    std::string header_text;
    std::vector<std::string> names;
    std::vector<int64_t> lengths;
    std::unordered_map<std::string, int> tid_of;
    std::vector<RefIndex> index;
    uint64_t first_record = 0;  // virtual offset just past the header
};

struct BamReader::Impl
{
    std::string path, index_path, reference;
    std::unique_ptr<Bgzf> bgzf;
    std::shared_ptr<const BamMeta> meta;
    RegionCursor cursor;
    std::vector<unsigned char> scratch;
    const unsigned char* record = nullptr;  // the bytes of the record readRecord parsed last (in the current block, or in `scratch`)

    void open();
    void parseHeader(BamMeta& m);
    void loadIndex(BamMeta& m);
    bool readRecord(BamRecord& rec);        // fixed fields + end_pos; valid until the next call
    void decodeText(BamRecord& rec) const;  // name / bases / quals of the record read last
    BamRecord lean_record;                  // getAlignLean's record (its text fields stay in `scratch`)
    RegionCursor query(int32_t tid, int64_t beg, int64_t end) const;
    bool next(RegionCursor& cur, BamRecord& rec);
};

namespace
{
bool fileExists(std::string const& p)
{
    std::ifstream f(p);
    return f.good();
}

void toRead(BamRecord const& rec, Read& read)
{
    read.set_fragment_id(rec.name);
    read.set_bases(rec.bases);
    read.set_quals(rec.quals);
    read.set_is_mapped((rec.flag & samflag::kUnmapped) == 0);
    read.set_is_first_mate((rec.flag & samflag::kFirstInPair) != 0);
    read.set_is_mate_mapped((rec.flag & samflag::kMateUnmapped) == 0);
    read.set_is_reverse_strand((rec.flag & samflag::kReverse) != 0);
    read.set_is_mate_reverse_strand((rec.flag & samflag::kMateReverse) != 0);
    read.set_chrom_id(rec.tid);
    read.set_pos(rec.pos);
    read.set_mapq(rec.mapq);
    read.set_mate_chrom_id(rec.mtid);
    read.set_mate_pos(rec.mpos);
}
}  // namespace

void BamReader::Impl::open()
{
    bgzf.reset(new Bgzf(path));
    static std::mutex cache_mutex;
    static std::map<std::pair<std::string, std::string>, std::weak_ptr<const BamMeta>> cache;
    std::lock_guard<std::mutex> lock(cache_mutex);
    auto& slot = cache[{ path, index_path }];
    meta = slot.lock();
    if (!meta)
    {
        auto fresh = std::make_shared<BamMeta>();
        parseHeader(*fresh);
        loadIndex(*fresh);
        meta = fresh;
        slot = meta;
    }
}

void BamReader::Impl::parseHeader(BamMeta& m)
{
    unsigned char b[8];
    bgzf->readExact(b, 4, "magic");
    if (memcmp(b, "BAM\1", 4) != 0)
    {
        if (memcmp(b, "CRAM", 4) == 0)
            throw std::runtime_error("ERROR: CRAM input is not supported by this reader: " + path);
        throw std::runtime_error("ERROR: Unknown alignment file format.");
    }
    bgzf->readExact(b, 4, "header length");
    if (le32(b) > (1u << 30))
        throw std::runtime_error("Corrupt BAM header (text length) in " + path);
    m.header_text.resize(le32(b));
    if (!m.header_text.empty())
        bgzf->readExact(&m.header_text[0], m.header_text.size(), "header text");
    bgzf->readExact(b, 4, "reference count");
    const uint32_t n_ref = le32(b);
    if (n_ref > (1u << 24))
        throw std::runtime_error("Corrupt BAM header (reference count) in " + path);
    for (uint32_t i = 0; i < n_ref; ++i)
    {
        bgzf->readExact(b, 4, "reference name length");
        if (le32(b) > (1u << 16))
            throw std::runtime_error("Corrupt BAM header (reference name length) in " + path);
        std::string name(le32(b), '\0');
        if (!name.empty())
            bgzf->readExact(&name[0], name.size(), "reference name");
        while (!name.empty() && name.back() == '\0')
            name.pop_back();
        bgzf->readExact(b, 4, "reference length");
        m.tid_of[name] = (int)i;
        m.names.push_back(name);
        m.lengths.push_back((int64_t)le32(b));
    }
    m.first_record = bgzf->tell();
}

void BamReader::Impl::loadIndex(BamMeta& m)
{
    std::string use = index_path;
    if (use.empty())
    {
        use = path + ".bai";
        if (!fileExists(use) && path.size() > 4 && path.compare(path.size() - 4, 4, ".bam") == 0)
            use = path.substr(0, path.size() - 4) + ".bai";
    }
    std::ifstream in(use, std::ios::binary);
    if (!in.good())
        throw std::runtime_error("ERROR: Failed to read index of " + path);
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t at = 0;
    auto need = [&](size_t n) {
        if (at + n > buf.size())
            throw std::runtime_error("ERROR: Truncated index " + use);
    };
    need(8);
    if (memcmp(buf.data(), "BAI\1", 4) != 0)
        throw std::runtime_error("ERROR: " + use + " is not a BAI index");
    const uint32_t n_ref = le32(buf.data() + 4);
    at = 8;
    if ((uint64_t)n_ref * 8 > buf.size())  // every reference has at least a bin count and an interval count
        throw std::runtime_error("ERROR: Corrupt index " + use);
    m.index.resize(n_ref);
    for (uint32_t r = 0; r < n_ref; ++r)
    {
        need(4);
        const uint32_t n_bin = le32(buf.data() + at);
        at += 4;
        for (uint32_t bi = 0; bi < n_bin; ++bi)
        {
            need(8);
            const uint32_t bin = le32(buf.data() + at);
            const uint32_t n_chunk = le32(buf.data() + at + 4);
            at += 8;
            need((size_t)n_chunk * 16);
            if (bin != 37450)  // the pseudo-bin holds statistics, not chunks
            {
                auto& chunks = m.index[r].bins[bin];
                for (uint32_t c = 0; c < n_chunk; ++c)
                    chunks.push_back(Chunk{ le64(buf.data() + at + c * 16), le64(buf.data() + at + c * 16 + 8) });
            }
            at += (size_t)n_chunk * 16;
        }
        need(4);
        const uint32_t n_intv = le32(buf.data() + at);
        at += 4;
        need((size_t)n_intv * 8);
        m.index[r].linear.resize(n_intv);
        for (uint32_t i = 0; i < n_intv; ++i)
            m.index[r].linear[i] = le64(buf.data() + at + (size_t)i * 8);
        at += (size_t)n_intv * 8;
    }
}

bool BamReader::Impl::readRecord(BamRecord& rec)
{
    // a record that lies inside the current block is used where it is; one that straddles blocks is put together in `scratch`
    unsigned char b4[4];
    const unsigned char* hdr = bgzf->take(4);
    if (!hdr)
    {
        const size_t got = bgzf->read(b4, 4);
        if (got == 0)
            return false;
        if (got != 4)
            throw std::runtime_error("Truncated BAM (record length) in " + path);
        hdr = b4;
    }
    const uint32_t block_size = le32(hdr);
    if (block_size < 32 || block_size > (1u << 29))
        throw std::runtime_error("Corrupt BAM record in " + path);
    const unsigned char* p = bgzf->take(block_size);
    if (!p)
    {
        scratch.resize(block_size);
        bgzf->readExact(scratch.data(), block_size, "record");
        p = scratch.data();
    }
    record = p;
    rec.tid = (int32_t)le32(p);
    rec.pos = (int32_t)le32(p + 4);
    const uint32_t l_name = p[8];
    rec.mapq = p[9];
    const uint32_t n_cigar = p[12] | (p[13] << 8);
    rec.flag = (uint16_t)(p[14] | (p[15] << 8));
    const uint32_t l_seq = le32(p + 16);
    rec.mtid = (int32_t)le32(p + 20);
    rec.mpos = (int32_t)le32(p + 24);
    size_t at = 32;
    if (at + l_name + (size_t)n_cigar * 4 + (l_seq + 1) / 2 + l_seq > block_size)
        throw std::runtime_error("Corrupt BAM record in " + path);
    rec.name_at = (uint32_t)at;
    rec.name_len = l_name ? l_name - 1 : 0;
    at += l_name;
    int64_t ref_span = 0;
    for (uint32_t c = 0; c < n_cigar; ++c)
    {
        const uint32_t op = le32(p + at + (size_t)c * 4);
        const uint32_t kind = op & 0xF;
        if (kind == 0 || kind == 2 || kind == 3 || kind == 7 || kind == 8)  // M D N = X consume the reference
            ref_span += op >> 4;
    }
    if ((rec.flag & 4) || n_cigar == 0)
        ref_span = 0;
    rec.end_pos = (int64_t)rec.pos + (ref_span > 0 ? ref_span : 1);
    at += (size_t)n_cigar * 4;
    rec.seq_at = (uint32_t)at;
    rec.seq_len = l_seq;
    return true;
}

void BamReader::Impl::decodeText(BamRecord& rec) const
{
    const unsigned char* p = record;
    rec.name.assign((const char*)p + rec.name_at, rec.name_len);
    const uint32_t l_seq = rec.seq_len;
    rec.bases.resize(l_seq);
    const unsigned char* packed = p + rec.seq_at;
    if (l_seq)
        unpackBases(packed, l_seq, &rec.bases[0]);
    const unsigned char* q = packed + (l_seq + 1) / 2;
    rec.quals.resize(l_seq);
    for (uint32_t i = 0; i < l_seq; ++i)
        rec.quals[i] = (char)(33 + q[i]);  // 0xFF ("no qualities") wraps like the uint8 -> char cast it mirrors
}

RegionCursor BamReader::Impl::query(int32_t tid, int64_t beg, int64_t end) const
{
    RegionCursor cur;
    if (tid < 0 || (size_t)tid >= meta->names.size())
        return cur;
    cur.valid = true;
    cur.tid = tid;
    cur.beg = std::max<int64_t>(beg, 0);
    cur.end = std::max(end, cur.beg);
    cur.finished = false;
    if ((size_t)tid >= meta->index.size() || cur.end <= cur.beg)
    {
        cur.finished = true;
        return cur;
    }
    RefIndex const& ri = meta->index[(size_t)tid];
    // records overlapping window beg >> 14 start at or after this offset; a 0 entry (empty window) just disables the cut
    uint64_t min_off = 0;
    if (!ri.linear.empty())
        min_off = ri.linear[std::min((size_t)(cur.beg >> 14), ri.linear.size() - 1)];
    const int64_t end_clamped = std::min<int64_t>(cur.end, (int64_t)1 << 29);
    std::vector<uint32_t> bins;
    reg2bins(std::min<int64_t>(cur.beg, ((int64_t)1 << 29) - 1), std::max<int64_t>(end_clamped, 1), bins);
    for (uint32_t bin : bins)
    {
        auto it = ri.bins.find(bin);
        if (it == ri.bins.end())
            continue;
        for (Chunk const& c : it->second)
        {
            if (c.end > min_off)
                cur.chunks.push_back(c);
        }
    }
    std::sort(cur.chunks.begin(), cur.chunks.end(), [](Chunk const& a, Chunk const& b) { return a.beg < b.beg; });
    std::vector<Chunk> merged;
    for (Chunk const& c : cur.chunks)
    {
        if (!merged.empty() && c.beg <= merged.back().end)
            merged.back().end = std::max(merged.back().end, c.end);
        else
            merged.push_back(c);
    }
    cur.chunks.swap(merged);
    if (cur.chunks.empty())
        cur.finished = true;
    return cur;
}

bool BamReader::Impl::next(RegionCursor& cur, BamRecord& rec)
{
    while (!cur.finished)
    {
        if (!cur.started)
        {
            cur.at = cur.chunks[cur.chunk].beg;
            cur.started = true;
        }
        if (cur.at >= cur.chunks[cur.chunk].end)
        {
            if (++cur.chunk == cur.chunks.size())
            {
                cur.finished = true;
                break;
            }
            cur.at = std::max(cur.at, cur.chunks[cur.chunk].beg);
            continue;
        }
        bgzf->seek(cur.at);
        if (!readRecord(rec))
        {
            cur.finished = true;
            break;
        }
        cur.at = bgzf->tell();
        if (rec.tid != cur.tid || (int64_t)rec.pos >= cur.end)
        {
            cur.finished = true;  // coordinate-sorted: nothing further can overlap
            break;
        }
        if (rec.end_pos > cur.beg)
            return true;
    }
    return false;
}

BamReader::BamReader(const std::string& path, const std::string& index_path, const std::string& reference) : impl_(new Impl)
{
    auto must_exist = [](std::string const& p) {
        if (!fileExists(p))
            throw std::runtime_error("ERROR: File " + p + " does not exist");
    };
    must_exist(path);
    if (!index_path.empty())
        must_exist(index_path);
    if (!reference.empty())
    {
        must_exist(reference);
        must_exist(reference + ".fai");
    }
    impl_->path = path;
    impl_->index_path = index_path;
    impl_->reference = reference;
    impl_->open();
}

BamReader::~BamReader() = default;
BamReader::BamReader(BamReader&&) noexcept = default;
BamReader& BamReader::operator=(BamReader&&) noexcept = default;
std::vector<std::string> const& BamReader::contigNames() const { return impl_->meta->names; }
std::vector<int64_t> const& BamReader::contigLengths() const { return impl_->meta->lengths; }
std::string const& BamReader::headerText() const { return impl_->meta->header_text; }

void BamReader::setRegion(const std::string& region_encoding)
{
    // hts_parse_reg: the text after the last ':' is "beg[-end]" with thousands separators allowed; a name that matches a
    // contig as a whole wins (contig names may hold ':')
    std::string name = region_encoding;
    int64_t beg = 0, end = (int64_t)1 << 29;
    auto whole = impl_->meta->tid_of.find(region_encoding);
    if (whole == impl_->meta->tid_of.end())
    {
        const size_t colon = region_encoding.rfind(':');
        if (colon != std::string::npos)
        {
            name = region_encoding.substr(0, colon);
            std::string range = region_encoding.substr(colon + 1);
            range.erase(std::remove(range.begin(), range.end(), ','), range.end());
            char* e = nullptr;
            const long long b = strtoll(range.c_str(), &e, 10);
            beg = b > 0 ? b - 1 : 0;
            if (*e == '-')
            {
                const long long en = strtoll(e + 1, &e, 10);
                end = en;
            }
            if (end < beg)
                end = beg;  // empty interval
        }
    }
    auto it = impl_->meta->tid_of.find(name);
    if (it == impl_->meta->tid_of.end())
        throw std::runtime_error("Failed to jump to " + region_encoding + " in " + impl_->path);
    impl_->cursor = impl_->query(it->second, beg, end);
}

bool BamReader::getAlign(Read& read)
{
    if (!impl_->cursor.valid)
        throw std::logic_error("Error: no region has been set on " + impl_->path);
    BamRecord rec;
    while (impl_->next(impl_->cursor, rec))
    {
        if (rec.flag & (samflag::kSupplementary | samflag::kSecondary))
            continue;
        impl_->decodeText(rec);
        toRead(rec, read);
        return true;
    }
    return false;
}

bool BamReader::getAlignLean(LeanAlign& out)
{
    if (!impl_->cursor.valid)
        throw std::logic_error("Error: no region has been set on " + impl_->path);
    BamRecord& rec = impl_->lean_record;
    while (impl_->next(impl_->cursor, rec))
    {
        if (rec.flag & (samflag::kSupplementary | samflag::kSecondary))
            continue;
        const unsigned char* p = impl_->record;
        out.name = (const char*)p + rec.name_at;
        out.name_len = rec.name_len;
        out.n_bases = rec.seq_len;
        out.packed_bases = p + rec.seq_at;
        out.text_bases = nullptr;
        out.chrom_id = rec.tid;
        out.pos = rec.pos;
        out.mate_chrom_id = rec.mtid;
        out.mate_pos = rec.mpos;
        out.is_mapped = (rec.flag & samflag::kUnmapped) == 0;
        out.is_first_mate = (rec.flag & samflag::kFirstInPair) != 0;
        out.is_mate_mapped = (rec.flag & samflag::kMateUnmapped) == 0;
        out.is_reverse_strand = (rec.flag & samflag::kReverse) != 0;
        out.is_mate_reverse_strand = (rec.flag & samflag::kMateReverse) != 0;
        return true;
    }
    return false;
}

bool ReadReader::getAlignLean(LeanAlign& out)
{
    if (!getAlign(lean_scratch_))
        return false;
    Read const& r = lean_scratch_;
    out.name = r.fragment_id().data();
    out.name_len = (uint32_t)r.fragment_id().size();
    out.n_bases = (uint32_t)r.bases().size();
    out.packed_bases = nullptr;
    out.text_bases = r.bases().data();
    out.chrom_id = r.chrom_id();
    out.pos = r.pos();
    out.mate_chrom_id = r.mate_chrom_id();
    out.mate_pos = r.mate_pos();
    out.is_mapped = r.is_mapped();
    out.is_first_mate = r.is_first_mate();
    out.is_mate_mapped = r.is_mate_mapped();
    out.is_reverse_strand = r.is_reverse_strand();
    out.is_mate_reverse_strand = r.is_mate_reverse_strand();
    return true;
}

void LeanAlign::appendBasesTo(std::string& out) const
{
    if (text_bases)
    {
        out.append(text_bases, n_bases);
        return;
    }
    const size_t at = out.size();
    out.resize(at + n_bases);
    if (n_bases)
        unpackBases(packed_bases, n_bases, &out[at]);
}

bool BamReader::getAlignedMate(const Read& read, Read& mate)
{
    const int32_t tid = read.is_mate_mapped() ? read.mate_chrom_id() : read.chrom_id();
    const int32_t beg = read.is_mate_mapped() ? read.mate_pos() : read.pos();
    RegionCursor cur = impl_->query(tid, beg, (int64_t)beg + 1);
    if (!cur.valid)
        return false;
    BamRecord rec;
    // like the original, `mate` is overwritten by every candidate looked at, also when none matches
    while (impl_->next(cur, rec))
    {
        impl_->decodeText(rec);
        toRead(rec, mate);
        if (mate.fragment_id() == read.fragment_id() && mate.is_first_mate() != read.is_first_mate())
            return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------------------------------
// read pairs and extraction
// ------------------------------------------------------------------------------------------------------------------
void ReadPairs::add(const Read& read)
{
    ReadPair& mates = pairs_[read.fragment_id()];
    const int before = mates.numInitialized();
    mates.add(read);
    num_reads_ += mates.numInitialized() - before;
}

const ReadPair& ReadPairs::operator[](const std::string& fragment_id) const
{
    auto it = pairs_.find(fragment_id);
    if (it == pairs_.end())
        throw std::runtime_error("Fragment " + fragment_id + " does not exist");
    return it->second;
}

void ReadPairs::getReads(std::vector<Read>& reads) const
{
    for (auto const& kv : pairs_)
    {
        if (kv.second.first_mate().is_initialized())
            reads.push_back(kv.second.first_mate());
        if (kv.second.second_mate().is_initialized())
            reads.push_back(kv.second.second_mate());
    }
}

void ReadPairs::getReads(std::vector<p_Read>& reads) const
{
    for (auto const& kv : pairs_)
    {
        if (kv.second.first_mate().is_initialized())
            reads.emplace_back(new Read(kv.second.first_mate()));
        if (kv.second.second_mate().is_initialized())
            reads.emplace_back(new Read(kv.second.second_mate()));
    }
}

void ReadPairs::takeReads(std::vector<p_Read>& reads)
{
    for (auto& kv : pairs_)
    {
        if (kv.second.first_mate().is_initialized())
            reads.emplace_back(new Read(std::move(kv.second.first_mate())));
        if (kv.second.second_mate().is_initialized())
            reads.emplace_back(new Read(std::move(kv.second.second_mate())));
    }
    clear();
}

void ReadPairs::clear()
{
    pairs_.clear();
    num_reads_ = 0;
}

bool isReadOrItsMateInRegion(Read& read, const Region& region)
{
    // the mate's extent is not known from this record; the read's own length stands in for it
    const int64_t len = (int64_t)read.bases().length();
    auto touches = [&](int64_t pos) { return !(pos > region.end || pos + len < region.start); };
    if (touches(read.pos()))
        return true;
    return read.chrom_id() == read.mate_chrom_id() && touches(read.mate_pos());
}

int extractMappedReadsFromRegion(ReadPairs& read_pairs, int max_num_reads, ReadReader& reader, const Region& region)
{
    Read read;
    unsigned total_length = 0, counted = 0;
    while (read_pairs.num_reads() != max_num_reads && reader.getAlign(read))
    {
        if (!read.bases().empty())
        {
            total_length += (unsigned)read.bases().length();
            ++counted;
        }
        if (isReadOrItsMateInRegion(read, region))
            read_pairs.add(read);
    }
    return counted ? (int)(total_length / counted) : 0;
}

void recoverMissingMates(ReadReader& reader, ReadPairs& read_pairs)
{
    // collected first: adding to the map while walking it is only safe because a recovered mate lands in an existing
    // entry, but collecting keeps that assumption out of the loop
    std::vector<Read> lonely;
    for (auto const& kv : read_pairs)
    {
        ReadPair const& pair = kv.second;
        if (pair.first_mate().is_initialized() && pair.second_mate().is_initialized())
            continue;
        Read const& have = pair.first_mate().is_initialized() ? pair.first_mate() : pair.second_mate();
        const int kMaxNormalDistanceBetweenMates = 1000;
        if (have.chrom_id() == have.mate_chrom_id() && std::abs(have.pos() - have.mate_pos()) < kMaxNormalDistanceBetweenMates)
            continue;  // the scan window would have held a mate this close
        lonely.push_back(have);
    }
    for (Read const& have : lonely)
    {
        Read mate;
        reader.getAlignedMate(have, mate);
        if (mate.is_initialized())
            read_pairs.add(mate);
    }
}

std::pair<int, int> extractReadsFromRegion(
    std::vector<p_Read>& all_reads, int max_num_reads, ReadReader& reader, const Region& region, unsigned longest_alt_insertion,
    int avr_fragment_length)
{
    reader.setRegion(region.getExtendedRegion((int64_t)avr_fragment_length * 3));
    ReadPairs read_pairs;
    const unsigned read_length = (unsigned)extractMappedReadsFromRegion(read_pairs, max_num_reads, reader, region);
    std::pair<int, int> extracted(read_pairs.num_reads(), 0);
    if (max_num_reads != read_pairs.num_reads() && read_length <= longest_alt_insertion * 2)
    {
        recoverMissingMates(reader, read_pairs);
        extracted.second = read_pairs.num_reads() - extracted.first;
    }
    read_pairs.takeReads(all_reads);
    return extracted;
}

void extractReads(
    ReadReader& reader, std::list<Region> const& target_regions, int max_num_reads, unsigned longest_alt_insertion,
    std::vector<p_Read>& all_reads, int avr_fragment_length)
{
    for (Region const& region : target_regions)
        extractReadsFromRegion(all_reads, max_num_reads, reader, region, longest_alt_insertion, avr_fragment_length);
}

void extractReads(
    const std::string& bam_path, const std::string& bam_index_path, const std::string& reference_path,
    std::list<Region> const& target_regions, int max_num_reads, unsigned longest_alt_insertion, std::vector<p_Read>& all_reads,
    int avr_fragment_length)
{
    BamReader reader(bam_path, bam_index_path, reference_path);
    extractReads(reader, target_regions, max_num_reads, longest_alt_insertion, all_reads, avr_fragment_length);
}

// ------------------------------------------------------------------------------------------------------------------
// extraction into the packed form
// ------------------------------------------------------------------------------------------------------------------
}  // namespace common

namespace paragraph
{
void PackedSite::clear()
{
    bases.clear();
    base_end.clear();
    fragment.clear();
    flags.clear();
    chrom_id.clear();
    pos.clear();
    mate_chrom_id.clear();
    mate_pos.clear();
}

namespace
{
// fragment ids of one site, kept once: the tables below hold views into it (blocks never move; reset() keeps them for the
// next site of the same thread)
class NameArena
{
public:
    std::string_view keep(const char* p, size_t n)
    {
        if (current_ == kNone || used_ + n > sizes_[current_])
        {
            const size_t next = current_ == kNone ? 0 : current_ + 1;
            if (next >= blocks_.size() || sizes_[next] < n)
            {
                blocks_.emplace(blocks_.begin() + (ptrdiff_t)next, new char[std::max<size_t>(kBlock, n)]);
                sizes_.insert(sizes_.begin() + (ptrdiff_t)next, std::max<size_t>(kBlock, n));
            }
            current_ = next;
            used_ = 0;
        }
        char* dst = blocks_[current_].get() + used_;
        memcpy(dst, p, n);
        used_ += n;
        return std::string_view(dst, n);
    }
    void reset()
    {
        current_ = kNone;
        used_ = 0;
    }

private:
    enum : size_t { kBlock = 16384, kNone = (size_t)-1 };
    std::vector<std::unique_ptr<char[]>> blocks_;
    std::vector<size_t> sizes_;
    size_t current_ = kNone, used_ = 0;
};

// fragment id -> value, open addressing over a flat entry list: no node per id, and clear() keeps the storage, so a worker
// thread allocates while its first sites grow the tables and not afterwards (std::unordered_map: one allocation per id, per site)
template <class V> class NameTable
{
public:
    struct Entry
    {
        std::string_view key;
        uint64_t hash;
        V value;
    };
    void clear()
    {
        entries_.clear();
        std::fill(index_.begin(), index_.end(), -1);
    }
    size_t size() const { return entries_.size(); }
    std::vector<Entry> const& entries() const { return entries_; }
    Entry& at(size_t i) { return entries_[i]; }
    // index of the entry of `key`, or -1
    int32_t find(std::string_view key, uint64_t h) const
    {
        if (index_.empty())
            return -1;
        const size_t mask = index_.size() - 1;
        for (size_t i = (size_t)h & mask;; i = (i + 1) & mask)
        {
            const int32_t e = index_[i];
            if (e < 0)
                return -1;
            if (entries_[(size_t)e].hash == h && entries_[(size_t)e].key == key)
                return e;
        }
    }
    // `key` must outlive the table's contents (a view into the arena)
    int32_t insert(std::string_view key, uint64_t h, V const& value)
    {
        if ((entries_.size() + 1) * 2 > index_.size())
            grow();
        entries_.push_back(Entry{ key, h, value });
        place((int32_t)entries_.size() - 1);
        return (int32_t)entries_.size() - 1;
    }
    static uint64_t hashOf(std::string_view s)
    {
        // 8 bytes at a time, multiply-fold (ids are 10-40 characters; the quality bar is "no clustering in a table half full")
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)s.size();
        const char* p = s.data();
        size_t n = s.size();
        while (n >= 8)
        {
            uint64_t w;
            memcpy(&w, p, 8);
            h = (h ^ w) * 0xFF51AFD7ED558CCDull;
            h ^= h >> 32;
            p += 8;
            n -= 8;
        }
        if (n)
        {
            uint64_t w = 0;
            memcpy(&w, p, n);
            h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
            h ^= h >> 29;
        }
        return h;
    }

private:
    void place(int32_t e)
    {
        const size_t mask = index_.size() - 1;
        size_t i = (size_t)entries_[(size_t)e].hash & mask;
        while (index_[i] >= 0)
            i = (i + 1) & mask;
        index_[i] = e;
    }
    void grow()
    {
        index_.assign(std::max<size_t>(256, index_.size() * 2), -1);
        for (int32_t e = 0; e < (int32_t)entries_.size(); ++e)
            place(e);
    }
    std::vector<Entry> entries_;
    std::vector<int32_t> index_;
};

// the reads of one target region while it is scanned: one slot pair per fragment id, like common::ReadPairs
struct RegionReads
{
    struct Kept
    {
        uint32_t base_begin, base_len;
        int32_t chrom_id, pos, mate_chrom_id, mate_pos;
        uint8_t flags;
    };
    typedef std::array<int32_t, 2> Slots;  // index into `kept` of first / second mate, -1 = empty
    NameTable<Slots> slots;
    std::vector<Kept> kept;
    std::string bases;
    std::vector<uint32_t> order;  // scratch of ordered()
    int num_reads = 0;

    void clear()
    {
        slots.clear();
        kept.clear();
        bases.clear();
        num_reads = 0;
    }
    void add(common::LeanAlign const& r, NameArena& names)
    {
        const std::string_view name(r.name, r.name_len);
        const uint64_t h = NameTable<Slots>::hashOf(name);
        int32_t e = slots.find(name, h);
        if (e < 0)
            e = slots.insert(names.keep(r.name, r.name_len), h, Slots{ -1, -1 });
        int32_t& slot = slots.at((size_t)e).value[r.is_first_mate ? 0 : 1];
        Kept k;
        k.base_begin = (uint32_t)bases.size();
        k.base_len = r.n_bases;
        r.appendBasesTo(bases);
        k.chrom_id = r.chrom_id;
        k.pos = r.pos;
        k.mate_chrom_id = r.mate_chrom_id;
        k.mate_pos = r.mate_pos;
        k.flags = (uint8_t)((r.is_reverse_strand ? PackedSite::REVERSE : 0) | (r.is_first_mate ? PackedSite::FIRST_MATE : 0)
                            | (r.is_mapped ? PackedSite::MAPPED : 0) | (r.is_mate_mapped ? PackedSite::MATE_MAPPED : 0)
                            | (r.is_mate_reverse_strand ? PackedSite::MATE_REVERSE : 0));
        if (slot < 0)
        {
            // an empty-sequence record does not make a slot "initialized" (Read::is_initialized), but a later one replaces it
            if (k.base_len > 0)
                ++num_reads;
            slot = (int32_t)kept.size();
            kept.push_back(k);
        }
        else
        {
            if (kept[(size_t)slot].base_len == 0 && k.base_len > 0)
                ++num_reads;
            else if (kept[(size_t)slot].base_len > 0 && k.base_len == 0)
                --num_reads;
            kept[(size_t)slot] = k;
        }
    }
    void add(common::Read const& r, NameArena& names)
    {
        common::LeanAlign l;
        l.name = r.fragment_id().data();
        l.name_len = (uint32_t)r.fragment_id().size();
        l.n_bases = (uint32_t)r.bases().size();
        l.text_bases = r.bases().data();
        l.chrom_id = r.chrom_id();
        l.pos = r.pos();
        l.mate_chrom_id = r.mate_chrom_id();
        l.mate_pos = r.mate_pos();
        l.is_mapped = r.is_mapped();
        l.is_first_mate = r.is_first_mate();
        l.is_mate_mapped = r.is_mate_mapped();
        l.is_reverse_strand = r.is_reverse_strand();
        l.is_mate_reverse_strand = r.is_mate_reverse_strand();
        add(l, names);
    }
    // entry indices in the order common::ReadPairs (a std::map by fragment id) walks them
    std::vector<uint32_t> const& ordered()
    {
        order.resize(slots.size());
        for (uint32_t i = 0; i < (uint32_t)order.size(); ++i)
            order[i] = i;
        auto const& e = slots.entries();
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return e[a].key < e[b].key; });
        return order;
    }
};

// what extractPacked needs besides the site it fills; one per worker thread, reused from site to site
struct ExtractScratch
{
    NameArena names;
    NameTable<uint32_t> fragment_of;  // across the regions of one site
    RegionReads pairs;
};

// common::isReadOrItsMateInRegion on a lean record
bool inRegion(common::LeanAlign const& r, common::Region const& region)
{
    const int64_t len = (int64_t)r.n_bases;
    auto touches = [&](int64_t pos) { return !(pos > region.end || pos + len < region.start); };
    if (touches(r.pos))
        return true;
    return r.chrom_id == r.mate_chrom_id && touches(r.mate_pos);
}
}  // namespace

void extractPacked(
    common::ReadReader& reader, std::list<common::Region> const& target_regions, int max_num_reads, unsigned longest_alt_insertion,
    PackedSite& site, int avr_fragment_length)
{
    if (!site.fragment.empty())
        throw std::logic_error("extractPacked: the site must be empty (fragment ids are assigned by name)");
    static thread_local ExtractScratch scratch;
    NameArena& names = scratch.names;
    NameTable<uint32_t>& fragment_of = scratch.fragment_of;
    RegionReads& pairs = scratch.pairs;
    names.reset();
    fragment_of.clear();
    common::LeanAlign rec;
    for (common::Region const& region : target_regions)
    {
        reader.setRegion(region.getExtendedRegion((int64_t)avr_fragment_length * 3));
        pairs.clear();
        unsigned total_length = 0, counted = 0;
        while (pairs.num_reads != max_num_reads && reader.getAlignLean(rec))
        {
            if (rec.n_bases)
            {
                total_length += rec.n_bases;
                ++counted;
            }
            if (inRegion(rec, region))
                pairs.add(rec, names);
        }
        const unsigned read_length = counted ? total_length / counted : 0;
        if (max_num_reads != pairs.num_reads && read_length <= longest_alt_insertion * 2)
        {
            // far-away mates of half-filled fragments (recoverMissingMates): rare, so a temporary Read per lookup is fine
            std::vector<common::Read> lonely;
            for (uint32_t entry : pairs.ordered())
            {
                auto const& kv = pairs.slots.entries()[entry];
                const int32_t a = kv.value[0], b = kv.value[1];
                const bool has_a = a >= 0 && pairs.kept[(size_t)a].base_len > 0, has_b = b >= 0 && pairs.kept[(size_t)b].base_len > 0;
                if (has_a == has_b)
                    continue;  // both there -- or neither, which is not a read at all
                RegionReads::Kept const& k = pairs.kept[(size_t)(has_a ? a : b)];
                if (k.chrom_id == k.mate_chrom_id && std::abs(k.pos - k.mate_pos) < 1000)
                    continue;
                common::Read have;
                have.setCoreInfo(std::string(kv.key), pairs.bases.substr(k.base_begin, k.base_len), "");
                have.set_is_first_mate((k.flags & PackedSite::FIRST_MATE) != 0);
                have.set_is_mate_mapped((k.flags & PackedSite::MATE_MAPPED) != 0);
                have.set_chrom_id(k.chrom_id);
                have.set_pos(k.pos);
                have.set_mate_chrom_id(k.mate_chrom_id);
                have.set_mate_pos(k.mate_pos);
                lonely.push_back(have);
            }
            for (common::Read const& have : lonely)
            {
                common::Read mate;
                reader.getAlignedMate(have, mate);
                if (mate.is_initialized())
                    pairs.add(mate, names);
            }
        }
        {
            const size_t more = pairs.kept.size();
            site.bases.reserve(site.bases.size() + pairs.bases.size());
            site.base_end.reserve(site.base_end.size() + more);
            site.fragment.reserve(site.fragment.size() + more);
            site.flags.reserve(site.flags.size() + more);
            site.chrom_id.reserve(site.chrom_id.size() + more);
            site.pos.reserve(site.pos.size() + more);
            site.mate_chrom_id.reserve(site.mate_chrom_id.size() + more);
            site.mate_pos.reserve(site.mate_pos.size() + more);
        }
        for (uint32_t entry : pairs.ordered())
        {
            auto const& kv = pairs.slots.entries()[entry];
            uint32_t fragment = 0;
            bool fragment_known = false;
            for (int32_t slot : kv.value)
            {
                if (slot < 0 || pairs.kept[(size_t)slot].base_len == 0)
                    continue;
                RegionReads::Kept const& k = pairs.kept[(size_t)slot];
                site.bases.append(pairs.bases, k.base_begin, k.base_len);
                site.base_end.push_back((uint32_t)site.bases.size());
                if (!fragment_known)
                {
                    int32_t f = fragment_of.find(kv.key, kv.hash);
                    if (f < 0)
                        f = fragment_of.insert(kv.key, kv.hash, (uint32_t)fragment_of.size());
                    fragment = fragment_of.at((size_t)f).value;
                    fragment_known = true;
                }
                site.fragment.push_back(fragment);
                site.flags.push_back(k.flags);
                site.chrom_id.push_back(k.chrom_id);
                site.pos.push_back(k.pos);
                site.mate_chrom_id.push_back(k.mate_chrom_id);
                site.mate_pos.push_back(k.mate_pos);
            }
        }
    }
}
}  // namespace paragraph

namespace common
{
// ------------------------------------------------------------------------------------------------------------------
// Read -> "alignments" entry
// ------------------------------------------------------------------------------------------------------------------
Json Read::toJson() const
{
    Json v = Json::object();
    auto text = [&](const char* key, std::string const& s) {
        if (!s.empty())
            v[key] = s;
    };
    auto number = [&](const char* key, int64_t n) {
        if (n)
            v[key] = n;
    };
    auto flag = [&](const char* key, bool b) {
        if (b)
            v[key] = true;
    };
    auto list = [&](const char* key, std::vector<std::string> const& items) {
        if (items.empty())
            return;
        Json arr = Json::array();
        for (auto const& s : items)
            arr.append(s);
        v[key] = arr;
    };
    text("fragmentId", fragment_id_);
    text("bases", bases_);
    text("quals", quals_);
    number("chromId", chrom_id_);
    number("pos", pos_);
    number("mapq", mapq_);
    flag("isReverseStrand", is_reverse_strand_);
    flag("isMateReverseStrand", is_mate_reverse_strand_);
    flag("isMapped", is_mapped_);
    flag("isFirstMate", is_first_mate_);
    flag("isMateMapped", is_mate_mapped_);
    number("mateChromId", mate_chrom_id_);
    number("matePos", mate_pos_);
    number("graphPos", graph_pos_);
    text("graphCigar", graph_cigar_);
    number("graphMapq", graph_mapq_);
    number("graphAlignmentScore", graph_alignment_score_);
    flag("isGraphAlignmentUnique", is_graph_alignment_unique_);
    flag("isGraphReverseStrand", is_graph_reverse_strand_);
    list("graphNodesSupported", nodes_);
    list("graphEdgesSupported", edges_);
    list("graphSequencesSupported", sequences_);
    if (status_ == BAD_ALIGN)
        v["graphMappingStatus"] = "BAD_ALIGN";
    else if (status_ == MAPPED)
        v["graphMappingStatus"] = "MAPPED";
    return v;
}
}  // namespace common
