"""Test-side FLAC ENCODER (there is no FLAC tool in the image): writes small but format-complete streams -- every subframe type,
Rice partitions with both parameter widths and escape partitions, wasted bits, all four channel assignments, 8/16/24-bit samples,
fixed and last-short block sizes, CRC-8 / CRC-16 -- for gomel_amd.flac to decode.  Independent of the decoder except for the CRC
polynomials (which the RFC 9639 known-answer file in test_flac.py pins)."""
import hashlib
import struct


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, bits):
        if bits == 0:
            return
        self.acc = (self.acc << bits) | (v & ((1 << bits) - 1))
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q):
        while q >= 32:
            self.put(0, 32); q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)


def _crc(data, poly, bits):
    c, top, mask = 0, 1 << (bits - 1), (1 << bits) - 1
    for b in data:
        c ^= b << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _rice(bw, res, order, blocksize, porder, method, escape_part=None):
    bw.put(method, 2)
    bw.put(porder, 4)
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    n_parts, i = 1 << porder, 0
    for part in range(n_parts):
        cnt = (blocksize >> porder) - (order if part == 0 else 0)
        vals = res[i:i + cnt]; i += cnt
        if escape_part is not None and part == escape_part:
            raw = max([1] + [(v if v >= 0 else ~v).bit_length() + 1 for v in vals])
            bw.put(esc, pbits); bw.put(raw, 5)
            for v in vals:
                bw.put(v, raw)
            continue
        folded = [(v << 1) if v >= 0 else ((-v << 1) - 1) for v in vals]
        mean = sum(folded) / max(1, len(folded))
        k = min(esc - 1, max(0, int(mean).bit_length() - 1))
        bw.put(k, pbits)
        for f in folded:
            bw.unary(f >> k)
            bw.put(f, k)


def subframe(bw, samples, bps, kind, order=0, coefs=None, shift=0, prec=12, porder=0, method=0, wasted=0, escape_part=None):
    n = len(samples)
    if wasted:
        assert all(s % (1 << wasted) == 0 for s in samples)
        samples = [s >> wasted for s in samples]
        bps -= wasted
    typ = {"constant": 0, "verbatim": 1, "fixed": 8 + order, "lpc": 31 + order}[kind]
    bw.put(0, 1); bw.put(typ, 6)
    if wasted:
        bw.put(1, 1); bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        bw.put(samples[0], bps)
    elif kind == "verbatim":
        for s in samples:
            bw.put(s, bps)
    else:
        if kind == "fixed":
            coefs, shift = FIXED[order], 0
        for s in samples[:order]:
            bw.put(s, bps)
        if kind == "lpc":
            bw.put(prec - 1, 4); bw.put(shift, 5)
            for c in coefs:
                bw.put(c, prec)
        res = []
        for i in range(order, n):
            pred = sum(c * samples[i - 1 - j] for j, c in enumerate(coefs)) >> shift
            res.append(samples[i] - pred)
        _rice(bw, res, order, n, porder, method, escape_part)


def frame(channels, bps, frame_no, mode="independent", specs=None, rate_code=9):
    """channels: list of equal-length integer lists (already left/right).  specs: one dict of subframe() keyword args per channel."""
    n, nch = len(channels[0]), len(channels)
    specs = specs or [dict(kind="verbatim")] * nch
    hdr = BitWriter()
    hdr.put(0x7FFC, 15); hdr.put(0, 1)
    table = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12}
    bs_code = table.get(n, 6 if n <= 256 else 7)
    hdr.put(bs_code, 4); hdr.put(rate_code, 4)
    ch_code = {"independent": nch - 1, "left_side": 8, "side_right": 9, "mid_side": 10}[mode]
    hdr.put(ch_code, 4); hdr.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps], 3); hdr.put(0, 1)
    if frame_no < 0x80:
        hdr.put(frame_no, 8)
    else:
        hdr.put(0xC0 | (frame_no >> 6), 8); hdr.put(0x80 | (frame_no & 0x3F), 8)
    if bs_code == 6:
        hdr.put(n - 1, 8)
    elif bs_code == 7:
        hdr.put(n - 1, 16)
    head = bytes(hdr.out)
    bw = BitWriter()
    bw.out = bytearray(head + bytes([_crc(head, 0x07, 8)]))
    if mode == "independent":
        coded, widths = channels, [bps] * nch
    else:
        l, r = channels
        side = [a - b for a, b in zip(l, r)]
        if mode == "left_side":
            coded, widths = [l, side], [bps, bps + 1]
        elif mode == "side_right":
            coded, widths = [side, r], [bps + 1, bps]
        else:
            coded, widths = [[(a + b) >> 1 for a, b in zip(l, r)], side], [bps, bps + 1]
    for s, w, sp in zip(coded, widths, specs):
        subframe(bw, s, w, **sp)
    bw.align()
    body = bytes(bw.out)
    return body + struct.pack(">H", _crc(body, 0x8005, 16))


def stream(frames, sample_rate, channels, bps, total, pcm_for_md5=b"", blocksize=4096, extra_metadata=True):
    x = (sample_rate << 44) | ((channels - 1) << 41) | ((bps - 1) << 36) | total
    si = struct.pack(">HH", blocksize, blocksize) + b"\x00\x00\x00" + b"\x00\x00\x00" + x.to_bytes(8, "big") + hashlib.md5(pcm_for_md5).digest()
    out = b"fLaC"
    if extra_metadata:
        out += bytes([0x00]) + (34).to_bytes(3, "big") + si
        out += bytes([0x84]) + (12).to_bytes(3, "big") + struct.pack("<I", 4) + b"test" + struct.pack("<I", 0)   # VORBIS_COMMENT, last block
    else:
        out += bytes([0x80]) + (34).to_bytes(3, "big") + si
    return out + b"".join(frames)
