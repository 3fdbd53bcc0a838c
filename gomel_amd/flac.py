"""FLAC stream decoder for `loadflac` (mel/impl.go:266-296, which calls github.com/mewkiz/flac): host-side file format code,
pure Python over the published format (RFC 9639): STREAMINFO, frame headers (CRC-8), CONSTANT / VERBATIM / FIXED / LPC subframes,
wasted bits, Rice-coded residuals (4- and 5-bit parameters, escape partitions), left/side, side/right and mid/side decorrelation,
CRC-16.  Returns integer samples per channel, frame by frame -- the shape the reference consumes (one subframe per channel)."""
import struct

import numpy as np


class FlacError(ValueError):
    pass


def _crc_table(poly, bits):
    top = 1 << (bits - 1)
    mask = (1 << bits) - 1
    t = []
    for b in range(256):
        c = b << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
        t.append(c)
    return t


_CRC8 = _crc_table(0x07, 8)
_CRC16 = _crc_table(0x8005, 16)


def crc8(data):
    c = 0
    for b in data:
        c = _CRC8[c ^ b]
    return c


def crc16(data):
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFF) ^ _CRC16[(c >> 8) ^ b]
    return c


class _Bits:
    """MSB-first bit reader over a bytes object."""

    def __init__(self, data, pos=0):
        self.d = data
        self.p = pos * 8          # bit position

    def read(self, n):
        if n == 0:
            return 0
        end = self.p + n
        if end > len(self.d) * 8:
            raise FlacError("unexpected end of stream")
        b0, b1 = self.p >> 3, (end + 7) >> 3
        v = int.from_bytes(self.d[b0:b1], "big")
        v >>= (b1 << 3) - end
        self.p = end
        return v & ((1 << n) - 1)

    def read_signed(self, n):
        v = self.read(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self):
        """Number of 0 bits before the next 1 bit (which is consumed)."""
        n = 0
        while True:
            byte = self.d[self.p >> 3] if (self.p >> 3) < len(self.d) else None
            if byte is None:
                raise FlacError("unexpected end of stream")
            rest = 8 - (self.p & 7)
            chunk = byte & ((1 << rest) - 1)
            if chunk == 0:
                n += rest
                self.p += rest
                continue
            lead = rest - chunk.bit_length()
            self.p += lead + 1
            return n + lead

    def align(self):
        self.p = (self.p + 7) & ~7

    @property
    def byte_pos(self):
        return self.p >> 3


_BLOCK = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608, 8: 256, 9: 512, 10: 1024, 11: 2048, 12: 4096, 13: 8192, 14: 16384, 15: 32768}
_RATE = {1: 88200, 2: 176400, 3: 192000, 4: 8000, 5: 16000, 6: 22050, 7: 24000, 8: 32000, 9: 44100, 10: 48000, 11: 96000}
_BPS = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}
_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _residual(br, order, blocksize, out):
    method = br.read(2)
    if method > 1:
        raise FlacError("reserved residual coding method")
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    porder = br.read(4)
    nparts = 1 << porder
    if blocksize % nparts or (blocksize >> porder) < order and porder:
        raise FlacError("bad Rice partition order")
    for part in range(nparts):
        count = (blocksize >> porder) - (order if part == 0 else 0)
        k = br.read(pbits)
        if k == esc:
            raw = br.read(5)
            for _ in range(count):
                out.append(br.read_signed(raw) if raw else 0)
        else:
            for _ in range(count):
                q = br.unary()
                v = (q << k) | br.read(k)
                out.append((v >> 1) ^ -(v & 1))


def _subframe(br, bps, blocksize):
    if br.read(1):
        raise FlacError("subframe padding bit set")
    typ = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
        bps -= wasted
        if bps <= 0:
            raise FlacError("wasted bits exceed the sample size")
    if typ == 0:                                            # CONSTANT
        s = [br.read_signed(bps)] * blocksize
    elif typ == 1:                                          # VERBATIM
        s = [br.read_signed(bps) for _ in range(blocksize)]
    elif 8 <= typ <= 12 or 32 <= typ <= 63:
        if typ <= 12:                                       # FIXED, order 0..4
            order = typ - 8
            coefs, shift = _FIXED[order], 0
            s = [br.read_signed(bps) for _ in range(order)]
        else:                                               # LPC, order 1..32
            order = typ - 31
            s = [br.read_signed(bps) for _ in range(order)]
            prec = br.read(4) + 1
            if prec == 16:
                raise FlacError("reserved LPC precision")
            shift = br.read_signed(5)
            if shift < 0:
                raise FlacError("negative LPC shift")
            coefs = [br.read_signed(prec) for _ in range(order)]
        if order > blocksize:
            raise FlacError("predictor order exceeds the block size")
        res = []
        _residual(br, order, blocksize, res)
        for r in res:                                       # s[n] = r + (sum_j c_j s[n-1-j]) >> shift
            acc = 0
            n = len(s)
            for j, c in enumerate(coefs):
                acc += c * s[n - 1 - j]
            s.append(r + (acc >> shift))
    else:
        raise FlacError("reserved subframe type %d" % typ)
    if wasted:
        s = [v << wasted for v in s]
    return s


def _utf8_number(br):
    b = br.read(8)
    if b < 0x80:
        return b
    n = 0
    while b & (0x80 >> n):
        n += 1
    if n < 2 or n > 7:
        raise FlacError("bad coded number")
    v = b & ((1 << (7 - n)) - 1)
    for _ in range(n - 1):
        c = br.read(8)
        if c & 0xC0 != 0x80:
            raise FlacError("bad coded number")
        v = (v << 6) | (c & 0x3F)
    return v


def decode(data):
    """bytes of a .flac file -> (info dict, frames) where frames is a list of per-frame lists of per-channel sample lists."""
    skip = 0
    if data[:3] == b"ID3" and len(data) >= 10:               # a leading ID3v2 tag (taggers prepend them; mewkiz/flac skips it): 10-byte
        size = 0                                             # header, 28-bit synch-safe length, optional 10-byte footer
        for c in data[6:10]:
            size = (size << 7) | (c & 0x7F)
        skip = 10 + size + (10 if data[5] & 0x10 else 0)
    if data[skip:skip + 4] != b"fLaC":
        raise FlacError("not a FLAC stream")
    pos, info = skip + 4, None
    while True:
        if pos + 4 > len(data):
            raise FlacError("truncated metadata")
        last, typ = data[pos] >> 7, data[pos] & 0x7F
        n = int.from_bytes(data[pos + 1:pos + 4], "big")
        body = data[pos + 4:pos + 4 + n]
        pos += 4 + n
        if typ == 0:
            if n < 34:
                raise FlacError("short STREAMINFO")
            x = int.from_bytes(body[10:18], "big")
            info = {"min_block": struct.unpack(">H", body[0:2])[0], "max_block": struct.unpack(">H", body[2:4])[0],
                    "sample_rate": x >> 44, "channels": ((x >> 41) & 7) + 1, "bits_per_sample": ((x >> 36) & 31) + 1,
                    "total_samples": x & ((1 << 36) - 1), "md5": bytes(body[18:34])}
        if last:
            break
    if info is None:
        raise FlacError("no STREAMINFO block")
    frames = []
    br = _Bits(data, pos)
    decoded = 0
    while br.byte_pos + 2 <= len(data):
        start = br.byte_pos
        if info["total_samples"] and decoded >= info["total_samples"]:
            break                                           # every sample STREAMINFO announces is in: what follows is not audio
        sync = br.read(15)
        if sync != 0x7FFC:
            if frames:
                break                                       # trailing bytes (an ID3v1 tag, padding): keep the frames decoded so far
            raise FlacError("lost frame sync at byte %d" % start)
        br.read(1)                                          # blocking strategy (only changes what the coded number counts)
        bs_code, sr_code = br.read(4), br.read(4)
        ch_code, bps_code = br.read(4), br.read(3)
        if br.read(1) or bs_code == 0 or sr_code == 15 or ch_code > 10 or bps_code == 3:
            raise FlacError("reserved value in a frame header")
        _utf8_number(br)
        if bs_code == 6:
            blocksize = br.read(8) + 1
        elif bs_code == 7:
            blocksize = br.read(16) + 1
        else:
            blocksize = _BLOCK[bs_code]
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        if crc8(data[start:br.byte_pos]) != br.read(8):
            raise FlacError("frame header CRC-8 mismatch at byte %d" % start)
        bps = _BPS[bps_code] if bps_code else info["bits_per_sample"]
        nch = ch_code + 1 if ch_code < 8 else 2
        extra = [0] * nch
        if ch_code == 8:
            extra[1] = 1                                    # left / side
        elif ch_code == 9:
            extra[0] = 1                                    # side / right
        elif ch_code == 10:
            extra[1] = 1                                    # mid / side
        sub = [_subframe(br, bps + extra[c], blocksize) for c in range(nch)]
        br.align()
        if crc16(data[start:br.byte_pos]) != br.read(16):
            raise FlacError("frame CRC-16 mismatch at byte %d" % start)
        if ch_code == 8:
            sub[1] = [l - s for l, s in zip(sub[0], sub[1])]
        elif ch_code == 9:
            sub[0] = [s + r for s, r in zip(sub[0], sub[1])]
        elif ch_code == 10:
            left, right = [], []
            for m, s in zip(sub[0], sub[1]):
                m = (m << 1) | (s & 1)
                left.append((m + s) >> 1)
                right.append((m - s) >> 1)
            sub = [left, right]
        frames.append(sub)
        decoded += blocksize
    return info, frames


def load(name):
    """The reference's loadflac (mel/impl.go:266-296): every sample of every subframe of every frame, FRAME BY FRAME (so a stereo
    file comes out as blocks of left samples followed by blocks of right samples -- the reference's `//break` is commented out),
    scaled by 1/65536 whatever the sample size.  Returns (float64 array, sample rate); raises FlacError / OSError."""
    with open(name, "rb") as f:
        data = f.read()
    info, frames = decode(data)
    out = [v for fr in frames for ch in fr for v in ch]
    return np.asarray(out, dtype=np.float64) / 65536.0, float(info["sample_rate"])
