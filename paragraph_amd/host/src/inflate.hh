// inflate.hh -- raw DEFLATE (RFC 1951) decoder for BGZF blocks: whole block in, whole block out, both sizes known.
//
// zlib's inflate() is a resumable state machine that can stop after any byte; a BGZF block (<= 64 KiB in, <= 64 KiB out, the
// inflated size in its trailer) needs none of that.  This decoder keeps 64 bits of input in a register, refills without a
// branch, decodes through two-level tables (11 bits for literals / lengths, 8 for distances; an entry carries the symbol's
// base value, its extra-bit count and the code length), emits up to three literals per refill, looks the NEXT symbol's entry
// up before it copies the current match (sixteen bytes at a time, two words per round, when the distance allows) while at
// least kFastIn / kFastOut bytes of slack remain, and falls back to a byte-exact loop near either end.  On the GPU box's host
// CPU: 3.6 GB/s of output on the end-to-end probe's BAM (97 % of its bytes come from matches of 44 bytes on average).
//
// Replaces the inflate half of htslib's bgzf_read_block as the reference's read extraction uses it
// (src/c++/lib/common/BamReader.cpp -> htslib bgzf.c); the CRC of every block is still checked by the caller.
// Checked against zlib on stored / fixed / dynamic blocks of every size class and on corrupted input
// (tests/host_cpp/test_hostio.cpp::testInflate).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace pginflate
{
enum : int
{
    kOk = 0,
    kBadData = 1,    // not a valid DEFLATE stream
    kShortOutput = 2 // the stream ends before / goes on after `out_len` bytes
};

namespace detail
{
enum : unsigned
{
    kLitBits = 11,
    kDistBits = 8,
    kMaxCodeLen = 15,
    kLitSyms = 288,
    kDistSyms = 32,
    // worst-case table sizes (primary + sub-tables): zlib's ENOUGH figures for 11 / 8 root bits are 2 * 1 << 11 at most here
    kLitTable = (1u << kLitBits) + 1400,
    kDistTable = (1u << kDistBits) + 400
};

// entry: bits 0-3 code length consumed by this level; bits 4-7 kind; bits 8-12 extra-bit count (or sub-table index bits);
// bits 16-31 base value / literal / sub-table start
enum : uint32_t
{
    kKindLiteral = 0u << 4,
    kKindLength = 1u << 4,   // also: distance entries
    kKindEnd = 2u << 4,
    kKindSub = 3u << 4,
    kKindBad = 4u << 4,
    kKindMask = 7u << 4
};

inline uint32_t entry(uint32_t value, uint32_t kind, uint32_t extra, uint32_t len) { return (value << 16) | (extra << 8) | kind | len; }

static const uint16_t kLengthBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t kLengthExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

inline uint32_t symbolEntry(bool is_dist, unsigned sym, unsigned len)
{
    if (is_dist)
        return sym < 30 ? entry(kDistBase[sym], kKindLength, kDistExtra[sym], len) : entry(0, kKindBad, 0, len);
    if (sym < 256)
        return entry(sym, kKindLiteral, 0, len);
    if (sym == 256)
        return entry(0, kKindEnd, 0, len);
    return sym < 286 ? entry(kLengthBase[sym - 257], kKindLength, kLengthExtra[sym - 257], len) : entry(0, kKindBad, 0, len);
}

inline unsigned reverseBits(unsigned code, unsigned len)
{
    // 16-bit reversal by bytes, then the top `len` bits
    static const struct Rev
    {
        unsigned char b[256];
        Rev()
        {
            for (unsigned i = 0; i < 256; ++i)
            {
                unsigned r = 0;
                for (unsigned k = 0; k < 8; ++k)
                    r |= ((i >> k) & 1u) << (7 - k);
                b[i] = (unsigned char)r;
            }
        }
    } rev;
    const unsigned r16 = ((unsigned)rev.b[code & 0xFF] << 8) | rev.b[(code >> 8) & 0xFF];
    return r16 >> (16 - len);
}

// Canonical Huffman code (RFC 1951 3.2.2) -> two-level lookup table indexed by the next `root` input bits (LSB first).
// Returns false for an over-subscribed code, and for an incomplete one unless it is the single-code case DEFLATE allows.
inline bool buildTable(const uint8_t* lens, unsigned n_syms, bool is_dist, unsigned root, uint32_t* table, unsigned table_cap)
{
    unsigned count[kMaxCodeLen + 1] = { 0 };
    for (unsigned s = 0; s < n_syms; ++s)
        ++count[lens[s]];
    count[0] = 0;
    unsigned used = 0;
    int left = 1;
    for (unsigned l = 1; l <= kMaxCodeLen; ++l)
    {
        left = left * 2 - (int)count[l];
        if (left < 0)
            return false;
        used += count[l];
    }
    const uint32_t bad = entry(0, kKindBad, 0, 1);
    if (left != 0 || used == 0)  // (a complete code covers every index of the primary table: nothing is left "bad")
        for (unsigned i = 0; i < (1u << root); ++i)
            table[i] = bad;
    if (used == 0)
        return true;  // no codes: fine as long as no symbol of this alphabet is ever decoded
    if (left > 0 && !(used == 1))
        return false;  // incomplete (only a single code may be incomplete: one distance code of length 1, RFC 1951 3.2.7)
    unsigned next_code[kMaxCodeLen + 2];
    {
        unsigned code = 0;
        for (unsigned l = 1; l <= kMaxCodeLen; ++l)
        {
            code = (code + count[l - 1]) << 1;
            next_code[l] = code;
        }
    }
    unsigned sub_next = 1u << root;
    // sub-tables: one per distinct `root`-bit prefix of the long codes; sized by the longest code under that prefix.  Long codes
    // come in canonical order, so all codes sharing a prefix are visited consecutively when symbols are walked by (length, symbol).
    // First pass over short codes fills the primary table directly; long codes are handled length by length below.
    unsigned code_of[kLitSyms];
    for (unsigned s = 0; s < n_syms; ++s)
        if (lens[s])
            code_of[s] = next_code[lens[s]]++;
    for (unsigned s = 0; s < n_syms; ++s)
    {
        const unsigned l = lens[s];
        if (l == 0 || l > root)
            continue;
        const unsigned r = reverseBits(code_of[s], l);
        const uint32_t e = symbolEntry(is_dist, s, l);
        for (unsigned i = r; i < (1u << root); i += 1u << l)
            table[i] = e;
    }
    // long codes: for every prefix find the longest length, allocate 2^(max - root) entries, fill
    bool any_long = false;
    for (unsigned l = root + 1; l <= kMaxCodeLen; ++l)
        any_long |= count[l] != 0;
    if (!any_long)
        return true;
    unsigned prefix_max[1u << kLitBits];
    for (unsigned i = 0; i < (1u << root); ++i)
        prefix_max[i] = 0;
    for (unsigned s = 0; s < n_syms; ++s)
    {
        const unsigned l = lens[s];
        if (l <= root)
            continue;
        const unsigned r = reverseBits(code_of[s], l) & ((1u << root) - 1);
        if (l > prefix_max[r])
            prefix_max[r] = l;
    }
    for (unsigned p = 0; p < (1u << root); ++p)
    {
        if (!prefix_max[p])
            continue;
        const unsigned sub_bits = prefix_max[p] - root;
        if (sub_next + (1u << sub_bits) > table_cap)
            return false;
        table[p] = entry(sub_next, kKindSub, sub_bits, root);
        for (unsigned i = 0; i < (1u << sub_bits); ++i)
            table[sub_next + i] = entry(0, kKindBad, 0, 1);
        // remember where the sub-table starts in prefix_max (reuse: high 16 bits = start)
        prefix_max[p] = (sub_next << 8) | sub_bits;
        sub_next += 1u << sub_bits;
    }
    for (unsigned s = 0; s < n_syms; ++s)
    {
        const unsigned l = lens[s];
        if (l <= root)
            continue;
        const unsigned r = reverseBits(code_of[s], l);
        const unsigned p = r & ((1u << root) - 1);
        const unsigned start = prefix_max[p] >> 8, sub_bits = prefix_max[p] & 0xFF;
        const uint32_t e = symbolEntry(is_dist, s, l - root);
        for (unsigned i = r >> root; i < (1u << sub_bits); i += 1u << (l - root))
            table[start + i] = e;
    }
    return true;
}

struct Tables
{
    uint32_t lit[kLitTable];
    uint32_t dist[kDistTable];
};

inline uint64_t load64(const unsigned char* p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v;  // little-endian hosts only (x86-64)
}
}  // namespace detail

// Inflates exactly one raw DEFLATE stream of `in_len` bytes into exactly `out_len` bytes.
// The caller guarantees 8 readable bytes after in + in_len and 16 writable bytes after out + out_len (slack for whole-word
// loads / stores; their contents are irrelevant).
inline int inflateBlock(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len)
{
    using namespace detail;
    static thread_local Tables T;
    static thread_local bool fixed_ready = false;
    static thread_local Tables F;  // the fixed code (RFC 1951 3.2.6), built once per thread

    const unsigned char* const in_end = in + in_len;
    unsigned char* const out_begin = out;
    unsigned char* const out_end = out + out_len;
    uint64_t bitbuf = 0;
    unsigned bitcnt = 0;
    size_t overrun = 0;  // whole bytes of zero "read" past in_end

    // refill to at least 56 bits; past the end of the input zeros are shifted in and counted
    auto refill = [&]() {
        if (in + 8 <= in_end + 8 && in <= in_end)  // the 8 bytes of slack make the load itself always safe while in <= in_end
        {
            if (in_end - in >= 8)
            {
                bitbuf |= load64(in) << bitcnt;
                in += (63 - bitcnt) >> 3;
                bitcnt |= 56;
                return;
            }
        }
        while (bitcnt <= 56)
        {
            if (in < in_end)
                bitbuf |= (uint64_t)*in++ << bitcnt;
            else
                ++overrun;
            bitcnt += 8;
        }
    };
    auto take = [&](unsigned n) -> uint32_t {
        const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1));
        bitbuf >>= n;
        bitcnt -= n;
        return v;
    };

    for (;;)
    {
        refill();
        const uint32_t final_block = take(1);
        const uint32_t type = take(2);
        const uint32_t* lit;
        const uint32_t* dist;
        if (type == 0)
        {
            // stored: skip to the byte boundary, LEN / NLEN, raw bytes
            take(bitcnt & 7);
            refill();
            const uint32_t len = take(16), nlen = take(16);
            if ((len ^ nlen) != 0xFFFFu)
                return kBadData;
            if (overrun * 8 > bitcnt)
                return kBadData;  // the header itself ran past the end of the input
            // give back the whole real bytes still in the bit buffer (the zeros counted in `overrun` sit above them)
            in -= bitcnt / 8 - overrun;
            overrun = 0;
            bitbuf = 0;
            bitcnt = 0;
            if ((size_t)(in_end - in) < len)
                return kBadData;
            if ((size_t)(out_end - out) < len)
                return kShortOutput;
            memcpy(out, in, len);
            in += len;
            out += len;
            if (final_block)
                break;
            continue;
        }
        if (type == 1)
        {
            if (!fixed_ready)
            {
                uint8_t lens[kLitSyms];
                for (unsigned s = 0; s < 144; ++s)
                    lens[s] = 8;
                for (unsigned s = 144; s < 256; ++s)
                    lens[s] = 9;
                for (unsigned s = 256; s < 280; ++s)
                    lens[s] = 7;
                for (unsigned s = 280; s < 288; ++s)
                    lens[s] = 8;
                uint8_t dl[kDistSyms];
                for (unsigned s = 0; s < 32; ++s)
                    dl[s] = 5;
                if (!buildTable(lens, 288, false, kLitBits, F.lit, kLitTable) || !buildTable(dl, 32, true, kDistBits, F.dist, kDistTable))
                    return kBadData;
                fixed_ready = true;
            }
            lit = F.lit;
            dist = F.dist;
        }
        else if (type == 2)
        {
            const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
            if (hlit > 286 || hdist > 30)
                return kBadData;
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            uint8_t cl[19] = { 0 };
            refill();
            for (uint32_t i = 0; i < hclen; ++i)
            {
                if (bitcnt < 3)
                    refill();
                cl[order[i]] = (uint8_t)take(3);
            }
            uint32_t cltab[1u << 7];
            {
                // code-length code: at most 7 bits, single level
                unsigned count[8] = { 0 };
                for (unsigned s = 0; s < 19; ++s)
                    ++count[cl[s]];
                count[0] = 0;
                int left = 1;
                unsigned used = 0;
                for (unsigned l = 1; l <= 7; ++l)
                {
                    left = left * 2 - (int)count[l];
                    if (left < 0)
                        return kBadData;
                    used += count[l];
                }
                if (used == 0 || (left > 0 && used != 1))
                    return kBadData;
                unsigned next_code[9], code = 0;
                for (unsigned l = 1; l <= 7; ++l)
                {
                    code = (code + count[l - 1]) << 1;
                    next_code[l] = code;
                }
                for (unsigned i = 0; i < 128; ++i)
                    cltab[i] = 0xFFFFFFFFu;
                for (unsigned s = 0; s < 19; ++s)
                {
                    const unsigned l = cl[s];
                    if (!l)
                        continue;
                    const unsigned r = reverseBits(next_code[l]++, l);
                    for (unsigned i = r; i < 128; i += 1u << l)
                        cltab[i] = (s << 8) | l;
                }
            }
            uint8_t lens[kLitSyms + kDistSyms];
            uint32_t n = 0;
            while (n < hlit + hdist)
            {
                refill();
                const uint32_t e = cltab[bitbuf & 127];
                if (e == 0xFFFFFFFFu)
                    return kBadData;
                take(e & 0xFF);
                const uint32_t sym = e >> 8;
                if (sym < 16)
                {
                    lens[n++] = (uint8_t)sym;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16)
                {
                    if (n == 0)
                        return kBadData;
                    val = lens[n - 1];
                    rep = 3 + take(2);
                }
                else if (sym == 17)
                    rep = 3 + take(3);
                else
                    rep = 11 + take(7);
                if (n + rep > hlit + hdist)
                    return kBadData;
                while (rep--)
                    lens[n++] = (uint8_t)val;
            }
            if (overrun * 8 > bitcnt)
                return kBadData;  // the code lengths ran past the end of the input
            if (lens[256] == 0)
                return kBadData;  // no end-of-block code
            uint8_t ll[kLitSyms] = { 0 }, dl[kDistSyms] = { 0 };
            memcpy(ll, lens, hlit);
            memcpy(dl, lens + hlit, hdist);
            if (!buildTable(ll, 288, false, kLitBits, T.lit, kLitTable) || !buildTable(dl, 32, true, kDistBits, T.dist, kDistTable))
                return kBadData;
            lit = T.lit;
            dist = T.dist;
        }
        else
            return kBadData;

        // ---- the block's symbols ------------------------------------------------------------------------------------
        enum : ptrdiff_t { kFastIn = 32, kFastOut = 320 };  // (258 + 31 bytes of copy overshoot + 3 literals)
        bool end_of_block = false;
        // fast loop: enough input that refills never meet the end, enough output room for two literals + the longest match
        // copied in 8-byte steps
        if (in_end - in >= kFastIn && out_end - out >= kFastOut)
        {
            // the entry of the NEXT symbol is looked up before the bytes of the current one are written: the table load does not
            // wait behind the copy
            bitbuf |= load64(in) << bitcnt;
            in += (63 - bitcnt) >> 3;
            bitcnt |= 56;
            uint32_t e = lit[bitbuf & ((1u << kLitBits) - 1)];
            for (;;)
            {
                if ((e & kKindMask) == kKindSub)
                {
                    bitbuf >>= kLitBits;
                    bitcnt -= kLitBits;
                    e = lit[(e >> 16) + (uint32_t)(bitbuf & ((1u << ((e >> 8) & 31)) - 1))];
                }
                bitbuf >>= e & 15;
                bitcnt -= e & 15;
                if ((e & kKindMask) == kKindLiteral)
                {
                    *out++ = (unsigned char)(e >> 16);
                    // a second and third literal without a refill: at most 15 + 15 + 15 bits were used of >= 56
                    e = lit[bitbuf & ((1u << kLitBits) - 1)];
                    if ((e & kKindMask) == kKindLiteral)
                    {
                        bitbuf >>= e & 15;
                        bitcnt -= e & 15;
                        *out++ = (unsigned char)(e >> 16);
                        e = lit[bitbuf & ((1u << kLitBits) - 1)];
                        if ((e & kKindMask) == kKindLiteral)
                        {
                            bitbuf >>= e & 15;
                            bitcnt -= e & 15;
                            *out++ = (unsigned char)(e >> 16);
                        }
                    }
                    if (!(in_end - in >= kFastIn && out_end - out >= kFastOut))
                        break;
                    bitbuf |= load64(in) << bitcnt;
                    in += (63 - bitcnt) >> 3;
                    bitcnt |= 56;
                    e = lit[bitbuf & ((1u << kLitBits) - 1)];
                    continue;
                }
                if ((e & kKindMask) != kKindLength)
                {
                    if ((e & kKindMask) == kKindEnd)
                    {
                        end_of_block = true;
                        break;
                    }
                    return kBadData;
                }
                // length (<= 15 + 5 bits so far), then the distance (<= 15 + 13): 48 of the >= 56 bits at most
                uint32_t length = (e >> 16) + (uint32_t)(bitbuf & ((1u << ((e >> 8) & 31)) - 1));
                bitbuf >>= (e >> 8) & 31;
                bitcnt -= (e >> 8) & 31;
                uint32_t d = dist[bitbuf & ((1u << kDistBits) - 1)];
                if ((d & kKindMask) == kKindSub)
                {
                    bitbuf >>= kDistBits;
                    bitcnt -= kDistBits;
                    d = dist[(d >> 16) + (uint32_t)(bitbuf & ((1u << ((d >> 8) & 31)) - 1))];
                }
                if ((d & kKindMask) != kKindLength)
                    return kBadData;
                bitbuf >>= d & 15;
                bitcnt -= d & 15;
                const uint32_t offset = (d >> 16) + (uint32_t)(bitbuf & ((1u << ((d >> 8) & 31)) - 1));
                bitbuf >>= (d >> 8) & 31;
                bitcnt -= (d >> 8) & 31;
                if (offset > (size_t)(out - out_begin))
                    return kBadData;
                const unsigned char* src = out - offset;
                unsigned char* dst = out;
                out += length;
                const bool more = in_end - in >= kFastIn && out_end - out >= kFastOut;
                if (more)
                {
                    bitbuf |= load64(in) << bitcnt;
                    in += (63 - bitcnt) >> 3;
                    bitcnt |= 56;
                    e = lit[bitbuf & ((1u << kLitBits) - 1)];
                }
                if (offset >= 16)
                {
                    do
                    {
                        memcpy(dst, src, 16);
                        memcpy(dst + 16, src + 16, 16);
                        dst += 32;
                        src += 32;
                    } while (dst < out);
                }
                else if (offset >= 8)
                {
                    do
                    {
                        memcpy(dst, src, 8);
                        dst += 8;
                        src += 8;
                    } while (dst < out);
                }
                else if (offset == 1)
                {
                    memset(dst, *src, length);
                }
                else
                {
                    do
                        *dst++ = *src++;
                    while (dst < out);
                }
                if (!more)
                    break;
            }
        }
        // careful loop: every byte of input and output checked
        while (!end_of_block)
        {
            refill();
            uint32_t e = lit[bitbuf & ((1u << kLitBits) - 1)];
            if ((e & kKindMask) == kKindSub)
            {
                take(kLitBits);
                e = lit[(e >> 16) + (uint32_t)(bitbuf & ((1u << ((e >> 8) & 31)) - 1))];
            }
            take(e & 15);
            const uint32_t kind = e & kKindMask;
            if (kind == kKindLiteral)
            {
                if (out == out_end)
                    return kShortOutput;
                *out++ = (unsigned char)(e >> 16);
                continue;
            }
            if (kind == kKindEnd)
                break;
            if (kind != kKindLength)
                return kBadData;
            const uint32_t length = (e >> 16) + take((e >> 8) & 31);
            refill();
            uint32_t d = dist[bitbuf & ((1u << kDistBits) - 1)];
            if ((d & kKindMask) == kKindSub)
            {
                take(kDistBits);
                d = dist[(d >> 16) + (uint32_t)(bitbuf & ((1u << ((d >> 8) & 31)) - 1))];
            }
            if ((d & kKindMask) != kKindLength)
                return kBadData;
            take(d & 15);
            const uint32_t offset = (d >> 16) + take((d >> 8) & 31);
            if (offset > (size_t)(out - out_begin))
                return kBadData;
            if ((size_t)(out_end - out) < length)
                return kShortOutput;
            const unsigned char* src = out - offset;
            for (uint32_t i = 0; i < length; ++i)
                out[i] = src[i];
            out += length;
        }
        if (overrun * 8 > bitcnt)
            return kBadData;  // symbols were decoded from bits that do not exist
        if (final_block)
            break;
    }
    if (overrun * 8 > bitcnt)
        return kBadData;
    return out == out_end ? kOk : kShortOutput;
}
}  // namespace pginflate
