"""A BAM + BAI writer that shares NO code with svision_amd/io/bam.py: a second reading of the SAM specification (SAMv1 4.2, 4.1
BGZF, 5.1.3 / 5.2 BAI), ``struct`` + ``zlib`` only, following htslib's conventions where the spec leaves a choice:

* BGZF blocks of at most 0xFF00 uncompressed bytes; ``policy="htslib"``: a record that does not fit into what is left of
  the block starts a new one (bgzf_flush_try), records larger than a block straddle several; ``policy="stream"``: blocks
  cut every 0xFF00 bytes wherever that falls (what re-compressing with bgzip gives) -- record headers straddle then too;
* the 28-byte EOF block; any deflate level (htslib's default is 6; 1 and 9 are in use);
* bin = reg2bin(pos, end) (5.3), chunks merged per run of equal bins, the pseudo-bin 37450 with the reference's offset range
  and its mapped / unmapped counts, the 16 kb linear index with empty windows back-filled from the right, n_no_coor;
* aux tags of every type (A c C s S i I f Z H and B arrays), a multi-KB MM:Z / ML:B,C pair, SEQ = * (l_seq 0), QUAL absent
  (0xFF), CIGARs of more than 65,535 operations as ``<l_seq>S<ref_len>N`` + CG:B,I (4.2.2), unmapped reads behind the
  last reference.

VERDICT r3 item 8 / weak 1: the product's readers had only ever read files written by the product's own writers."""
import struct
import zlib

OPS = "MIDNSHP=X"
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
BLOCK = 0xFF00


def reg2bin(beg, end):
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


def ref_len(cigar):
    return sum(n for n, op in cigar if op in "MDN=X")


def query_len(cigar):
    return sum(n for n, op in cigar if op in "MIS=X")


def _tag(name, typ, value):
    out = name.encode() + typ[:1].encode()
    if typ in "AcCsSiIf":
        return out + struct.pack("<" + {"A": "c", "c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[typ],
                                 value.encode() if typ == "A" else value)
    if typ in "ZH":
        return out + value.encode() + b"\x00"
    sub = typ[1]                                                # "Bc", "BC", "Bs", "BS", "Bi", "BI", "Bf"
    return out + sub.encode() + struct.pack("<i", len(value)) + struct.pack("<%d%s" % (len(value), {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]), *value)


def encode_record(rec):
    """rec: dict(tid, pos, qname, flag, mapq, cigar=[(n, op)], seq=str or '*', qual=bytes or None, tags=[(name, type, value)],
    next_tid=-1, next_pos=-1, tlen=0) -> the record's bytes including its block_size."""
    cigar = list(rec.get("cigar", []))
    tags = list(rec.get("tags", []))
    seq = rec.get("seq", "*")
    l_seq = 0 if seq == "*" else len(seq)
    words = [n << 4 | OPS.index(op) for n, op in cigar]
    end = rec["pos"] + (ref_len(cigar) or 1)
    if len(words) > 65535:                                      # 4.2.2: the real CIGAR moves to CG:B,I
        tags.append(("CG", "BI", words))
        words = [l_seq << 4 | 4, ref_len(cigar) << 4 | 3]
    name = rec["qname"].encode() + b"\x00"
    if seq == "*":
        packed = b""
    else:
        codes = ["=ACMGRSVTWYHKDBN".index(c) for c in seq]
        if len(codes) & 1:
            codes.append(0)
        packed = bytes(codes[i] << 4 | codes[i + 1] for i in range(0, len(codes), 2))
    qual = rec.get("qual")
    qual = b"\xff" * l_seq if qual is None else bytes(qual)
    assert len(qual) == l_seq
    body = struct.pack("<iiBBHHHiiii", rec["tid"], rec["pos"], len(name), rec["mapq"], reg2bin(rec["pos"], end) if rec["tid"] >= 0 else 4680,
                       len(words), rec["flag"], l_seq, rec.get("next_tid", -1), rec.get("next_pos", -1), rec.get("tlen", 0))
    body += name + struct.pack("<%dI" % len(words), *words) + packed + qual + b"".join(_tag(*t) for t in tags)
    return struct.pack("<i", len(body)) + body


def _bgzf_block(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    c = co.compress(data) + co.flush()
    assert len(c) + 26 <= 65536
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(c) + 25) + c
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def write_bam(path, references, records, level=6, policy="htslib", header_text=None, index=True):
    """references: [(name, length)]; records: coordinate-sorted dicts (encode_record), unmapped ones (tid -1) last.
    Writes path (+ path.bai) and returns the virtual offset of every record."""
    if header_text is None:
        header_text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in references) + \
                      "@RG\tID:rg1\tSM:sampleA\n@RG\tID:rg2\tSM:sampleB\n@PG\tID:htslike\tPN:htslike\tVN:1\n"
    head = b"BAM\x01" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", len(references))
    for name, length in references:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\x00" + struct.pack("<i", length)
    blocks, cur = [], bytearray()
    starts = []                                                 # per record: (block index, offset in the block)

    def put(data, whole):
        nonlocal cur
        if policy == "htslib" and whole and len(cur) + len(data) > BLOCK and cur:
            blocks.append(bytes(cur))                           # bgzf_flush_try: the record starts a fresh block
            cur = bytearray()
        at = (len(blocks), len(cur))
        p = 0
        while p < len(data):
            room = BLOCK - len(cur)
            cur += data[p:p + room]
            p += room
            if len(cur) == BLOCK:
                blocks.append(bytes(cur))
                cur = bytearray()
        return at
    put(head, False)
    if policy == "htslib":                                      # samtools flushes the header into blocks of its own
        blocks.append(bytes(cur))
        cur = bytearray()
    for rec in records:
        starts.append(put(encode_record(rec), True))
    end_at = (len(blocks), len(cur))
    if cur:
        blocks.append(bytes(cur))
    coff, p, out = [], 0, []
    for b in blocks:
        coff.append(p)
        z = _bgzf_block(b, level)
        out.append(z)
        p += len(z)
    coff.append(p)                                              # where the EOF block starts
    with open(path, "wb") as f:
        f.write(b"".join(out) + EOF_BLOCK)

    def voff(at):
        bi, off = at
        if bi < len(blocks) and off == len(blocks[bi]) and off:  # exactly behind a block's last byte: the next block's first
            bi, off = bi + 1, 0
        return coff[bi] << 16 | off
    voffs = [voff(a) for a in starts] + [voff(end_at)]
    if index:
        write_bai(path + ".bai", references, records, voffs)
    return voffs


def write_bai(path, references, records, voffs):
    """htslib's hts_idx_push / hts_idx_finish for a coordinate-sorted record list."""
    n_ref = len(references)
    bins = [dict() for _ in range(n_ref)]                       # bin -> [(beg, end)]
    linear = [dict() for _ in range(n_ref)]
    meta = [[None, None, 0, 0] for _ in range(n_ref)]           # off_beg, off_end, n_mapped, n_unmapped
    n_no_coor = 0
    last_tid, last_bin, save_off = None, None, None
    for i, rec in enumerate(records):
        tid, beg_v, end_v = rec["tid"], voffs[i], voffs[i + 1]
        if tid < 0:
            n_no_coor += 1
            continue
        beg = rec["pos"]
        end = beg + (ref_len(rec.get("cigar", [])) or 1)
        b = reg2bin(beg, end)
        if tid != last_tid or b != last_bin:
            if last_tid is not None and last_bin is not None:
                bins[last_tid].setdefault(last_bin, []).append((save_off, beg_v))
            last_tid, last_bin, save_off = tid, b, beg_v
        m = meta[tid]
        if m[0] is None:
            m[0] = beg_v
        m[1] = end_v
        m[3 if rec["flag"] & 4 else 2] += 1
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            linear[tid].setdefault(w, beg_v)
    if last_tid is not None:
        bins[last_tid].setdefault(last_bin, []).append((save_off, voffs[len([r for r in records if r["tid"] >= 0])]))
    out = b"BAI\x01" + struct.pack("<i", n_ref)
    for t in range(n_ref):
        bl = bins[t]
        has = meta[t][0] is not None
        out += struct.pack("<i", len(bl) + (1 if has else 0))
        for b in sorted(bl):
            out += struct.pack("<Ii", b, len(bl[b]))
            for beg, end in bl[b]:
                out += struct.pack("<QQ", beg, end)
        if has:
            out += struct.pack("<IiQQQQ", 37450, 2, meta[t][0], meta[t][1], meta[t][2], meta[t][3])
        n_intv = (max(linear[t]) + 1) if linear[t] else 0
        offs = [linear[t].get(w) for w in range(n_intv)]
        for w in range(n_intv - 2, -1, -1):                     # empty windows take the offset of the next one (hts_idx_finish)
            if offs[w] is None:
                offs[w] = offs[w + 1]
        out += struct.pack("<i", n_intv) + struct.pack("<%dQ" % n_intv, *offs)
    out += struct.pack("<Q", n_no_coor)
    with open(path, "wb") as f:
        f.write(out)
