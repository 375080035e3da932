"""Minimal coordinate-sorted BAM + BAI writer for synthetic test / probe inputs (SAM/BAM specification v1, sections 4.1-4.2
and 5.2; BGZF blocks via zlib).  Test infrastructure: the product only READS BAMs (paragraph_amd/host/src/io.cpp).

    write_bam(path, contigs=[("chr1", 100000)], records=[dict(name=, tid=, pos=, seq=, qual=, flag=, mapq=, mtid=, mpos=, cigar=[(len, op)])])

Records must be sorted by (tid, pos).  cigar defaults to <len(seq)>M.
"""
import struct
import zlib

_OPS = "MIDNSHP=X"
_SEQ = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
BLOCK = 0xff00


def _bgzf_block(payload):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(payload) + comp.flush()
    bsize = len(cdata) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + cdata
            + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _encode(rec):
    seq = rec["seq"]
    cigar = rec.get("cigar") or [(len(seq), "M")]
    flag = rec.get("flag", 0)
    ref_span = sum(n for n, op in cigar if op in "MDN=X")
    end = rec["pos"] + (ref_span if ref_span and not flag & 4 else 1)
    name = rec["name"].encode() + b"\0"
    packed = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
        packed[i // 2] |= _SEQ.get(c, 15) << (0 if i & 1 else 4)
    qual = bytes(ord(q) - 33 for q in rec.get("qual") or "I" * len(seq))
    body = struct.pack("<iiBBHHHiiii", rec["tid"], rec["pos"], len(name), rec.get("mapq", 60), _reg2bin(rec["pos"], end), len(cigar), flag,
                       len(seq), rec.get("mtid", -1), rec.get("mpos", -1), rec.get("tlen", 0))
    body += name + b"".join(struct.pack("<I", n << 4 | _OPS.index(op)) for n, op in cigar) + bytes(packed) + qual
    return struct.pack("<i", len(body)) + body, end


def write_bam(path, contigs, records, header_text=None):
    text = (header_text or "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs)).encode()
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(contigs))
    for name, length in contigs:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length)
    out = bytearray()
    buf = bytearray(head)
    out += _bgzf_block(bytes(buf))  # header in its own block, like samtools
    buf = bytearray()
    block_start = len(out)
    bins = [dict() for _ in contigs]      # tid -> bin -> [[beg, end]]
    linear = [dict() for _ in contigs]    # tid -> window -> min voffset
    last = (-1, -1)

    def flush():
        nonlocal buf, block_start
        if buf:
            out.extend(_bgzf_block(bytes(buf)))
            buf = bytearray()
        block_start = len(out)

    pending = []  # (tid, bin, beg_voffset, windows) waiting for their end voffset
    for rec in records:
        assert (rec["tid"], rec["pos"]) >= last, "records must be coordinate-sorted"
        last = (rec["tid"], rec["pos"])
        data, end = _encode(rec)
        if len(buf) + len(data) > BLOCK:
            flush()
        beg_v = block_start << 16 | len(buf)
        buf += data
        end_v = block_start << 16 | len(buf)
        if len(buf) >= BLOCK:
            flush()
            end_v = block_start << 16
        if rec["tid"] >= 0:
            chunks = bins[rec["tid"]].setdefault(_reg2bin(rec["pos"], end), [])
            if chunks and chunks[-1][1] >= beg_v:
                chunks[-1][1] = end_v
            else:
                chunks.append([beg_v, end_v])
            for w in range(rec["pos"] >> 14, ((end - 1) >> 14) + 1):
                linear[rec["tid"]].setdefault(w, beg_v)
    flush()
    out += _bgzf_block(b"")  # EOF marker
    with open(path, "wb") as f:
        f.write(out)
    bai = bytearray(b"BAI\1" + struct.pack("<i", len(contigs)))
    for tid in range(len(contigs)):
        bai += struct.pack("<i", len(bins[tid]))
        for b, chunks in sorted(bins[tid].items()):
            bai += struct.pack("<Ii", b, len(chunks))
            for beg, end in chunks:
                bai += struct.pack("<QQ", beg, end)
        n_intv = max(linear[tid]) + 1 if linear[tid] else 0
        bai += struct.pack("<i", n_intv)
        prev = 0
        for w in range(n_intv):
            prev = linear[tid].get(w, prev)
            bai += struct.pack("<Q", prev)
    with open(path + ".bai", "wb") as f:
        f.write(bai)


def write_fasta(path, contigs):
    """contigs: [(name, sequence)], 60 bases per line, plus the .fai."""
    with open(path, "w") as f, open(path + ".fai", "w") as fai:
        at = 0
        for name, seq in contigs:
            head = ">%s\n" % name
            f.write(head)
            at += len(head)
            fai.write("%s\t%d\t%d\t60\t61\n" % (name, len(seq), at))
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
            at += len(seq) + (len(seq) + 59) // 60
