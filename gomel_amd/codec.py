"""Host-side wire formats of the reference's mel package (mel/impl.go): the PNG spectrogram image with its float16
metadata, and mono 16-bit WAV.  These stay on the host side of the C-ABI boundary (in the Go drop-in they remain the
reference's own Go code); this Python mirror exists so that `Mel.ToWavPng` / `Mel.ToMelWav` run end to end.

PNG: 8-bit R,G carry the two spectrogram channels scaled by the GLOBAL min/max, x = time, y = mel band (flipped when
YReverse); four little-endian float16 {max, min, samples per column, sample rate} sit in the blue channel of column 0
at y >= H-8 (mel/impl.go:127-193 writer, :52-118 reader).
"""
import struct
import zlib

import numpy as np


# ----------------------------------------------------------------------------- float16 metadata
def packFloat16ToBytes(f):
    """mel/impl.go:120-125: float16.Fromfloat32(float32(f)), little endian."""
    return np.array([np.float32(f)], dtype=np.float32).astype("<f2").tobytes()


def unpackBytesToFloat64(b):
    """mel/impl.go:46-50."""
    return float(np.frombuffer(bytes(b[:2]), dtype="<f2")[0])


# ----------------------------------------------------------------------------- PNG container (8/16-bit gray, RGB, RGBA, palette; no interlace)
def read_png(path):
    """Decode to an [H, W, 4] uint8 RGBA array holding what Go's color.RGBA()>>8 yields for each pixel."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, plte, trns = 8, [], None, None
    W = H = depth = ctype = interlace = None
    while pos + 8 <= len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            W, H, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif tag == b"tRNS":
            trns = body
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    if W is None or interlace:
        raise ValueError("unsupported PNG (missing IHDR or interlaced)")
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    if depth not in (8, 16) and not (ctype in (0, 3) and depth in (1, 2, 4)):
        raise ValueError("unsupported PNG bit depth")
    raw = zlib.decompress(b"".join(idat))
    bpp = max(1, chans * depth // 8)
    stride = (W * chans * depth + 7) // 8
    rows = np.zeros((H, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    o = 0
    for y in range(H):
        ft = raw[o]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=o + 1).copy()
        o += 1 + stride
        if ft == 1:
            for i in range(bpp, stride):
                line[i] = (int(line[i]) + int(line[i - bpp])) & 255
        elif ft == 2:
            line = (line.astype(np.uint16) + prev).astype(np.uint8)
        elif ft == 3:
            for i in range(stride):
                left = int(line[i - bpp]) if i >= bpp else 0
                line[i] = (int(line[i]) + ((left + int(prev[i])) >> 1)) & 255
        elif ft == 4:
            for i in range(stride):
                a = int(line[i - bpp]) if i >= bpp else 0
                b = int(prev[i])
                c = int(prev[i - bpp]) if i >= bpp else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                line[i] = (int(line[i]) + pr) & 255
        elif ft != 0:
            raise ValueError("bad PNG filter")
        rows[y] = line
        prev = line
    if depth < 8:
        bits = np.unpackbits(rows, axis=1)[:, :W * depth].reshape(H, W, depth)
        samp = bits.dot(1 << np.arange(depth - 1, -1, -1)).astype(np.uint16)[:, :, None]
        if ctype == 0:
            samp = samp * (255 // ((1 << depth) - 1))
        s8 = samp.astype(np.uint8)
    elif depth == 16:
        s8 = rows.reshape(H, W, chans, 2)[:, :, :, 0]  # top byte == RGBA()>>8 for 16-bit samples
    else:
        s8 = rows.reshape(H, W, chans)
    out = np.zeros((H, W, 4), dtype=np.uint8)
    out[:, :, 3] = 255
    if ctype == 0:
        out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = s8[:, :, 0]
    elif ctype == 2:
        out[:, :, :3] = s8
    elif ctype == 3:
        out[:, :, :3] = plte[s8[:, :, 0]]
        if trns is not None:
            a = np.full(256, 255, dtype=np.uint8)
            a[:len(trns)] = np.frombuffer(trns, dtype=np.uint8)
            out[:, :, 3] = a[s8[:, :, 0]]
    elif ctype == 4:
        out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = s8[:, :, 0]
        out[:, :, 3] = s8[:, :, 1]
    else:
        out[:, :, :] = s8
    if ctype in (4, 6) or (ctype == 3 and trns is not None):
        # Go's color.NRGBA.RGBA() premultiplies by alpha; with A = 255 (what dumpimage writes) this is the identity
        a = out[:, :, 3:4].astype(np.uint32)
        rgb = out[:, :, :3].astype(np.uint32)
        out[:, :, :3] = ((rgb * 0x101 * (a * 0x101) // 0xFFFF) >> 8).astype(np.uint8)
    return out


def write_png(path, img):
    """Encode an [H, W, 4] uint8 array as 8-bit RGBA (what image.NewNRGBA + png.Encode produce, modulo compression)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, C = img.shape
    if C != 4:
        raise ValueError("RGBA expected")
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(H))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ----------------------------------------------------------------------------- spectrogram image
def loadpng(name, reverse):
    """mel/impl.go:52-118.  Returns (buf [W*H, 2] float64 x-major, samples, samplerate); ([], 0, 0) if the file cannot
    be read or decoded (the reference prints the error and returns nil)."""
    try:
        img = read_png(name)
    except (OSError, ValueError, KeyError, zlib.error, struct.error) as e:
        print(str(e))
        return np.zeros((0, 2)), 0.0, 0.0
    H, W = img.shape[0], img.shape[1]
    if H < 8:
        raise IndexError("image height < 8: metadata slice out of range (reference panics at impl.go:100)")
    px = img[::-1] if reverse else img                      # px[y] is the row the reference reads at loop index y
    floats = bytes(int(v) for v in px[H - 8:H, 0, 2])       # blue @ x == 0, y >= mels-8
    mgc_max, mgc_min = unpackBytesToFloat64(floats[0:2]), unpackBytesToFloat64(floats[2:4])
    samples_in_mel, sr = unpackBytesToFloat64(floats[4:6]), unpackBytesToFloat64(floats[6:8])
    if mgc_max == samples_in_mel:
        samples_in_mel = 0.0
    v = np.stack([px[:, :, 0].T, px[:, :, 1].T], axis=2).astype(np.float64) / 255.0   # [W, H, 2]: x outer, y inner
    buf = np.ascontiguousarray((v * (mgc_max - mgc_min) + mgc_min).reshape(W * H, 2))
    return buf, samples_in_mel * float(W), sr


def dumpimage(name, buf, mels, reverse, samples_in_mel, sr):
    """mel/impl.go:127-193."""
    buf = np.asarray(buf, dtype=np.float64)
    stride = len(buf) // mels
    b = buf[:stride * mels]
    mgc_max, mgc_min = float(b.max()), float(b.min())      # the scan at impl.go:140-152 visits every element
    floats = packFloat16ToBytes(mgc_max) + packFloat16ToBytes(mgc_min) + packFloat16ToBytes(samples_in_mel) + packFloat16ToBytes(sr)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = (b.reshape(stride, mels, 2) - mgc_min) / (mgc_max - mgc_min)
    q = (255 * v).astype(np.int64).astype(np.uint8)        # uint8(int(255*val)): truncation
    img = np.zeros((mels, stride, 4), dtype=np.uint8)
    meta_start = mels - len(floats)
    for y in range(mels):
        row = (mels - y - 1) if reverse else y
        img[row, :, 0] = q[:, y, 0]
        img[row, :, 1] = q[:, y, 1]
        img[row, :, 3] = 255
        if y >= meta_start:
            img[row, 0, 2] = floats[y - meta_start]
    write_png(name, img)
    return None


def dumpbuffer(buf, mels):
    """mel/impl.go:16-44 ((*Mel).Image): per-channel min/max, uint16 = R | G << 8."""
    buf = np.asarray(buf, dtype=np.float64)
    stride = len(buf) // mels
    b = buf[:stride * mels]
    mx, mn = b.max(axis=0), b.min(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = (b - mn) / (mx - mn)
    q = (255 * v).astype(np.int64)
    return (q[:, 0].astype(np.uint16) | (q[:, 1].astype(np.uint16) << 8)).astype(np.uint16)


# ----------------------------------------------------------------------------- WAV
def dumpwav(name, data, sr):
    """mel/impl.go:195-232: mono, 16-bit PCM through beep's wav.Encode (samples clamped to [-1, 1], scaled by 2^15-1)."""
    v = np.clip(np.asarray(data, dtype=np.float64), -1.0, 1.0)
    pcm = (v * 32767.0).astype("<i2").tobytes()
    sr = int(sr)
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16)
           + b"data" + struct.pack("<I", len(pcm)))
    with open(name, "wb") as f:
        f.write(hdr + pcm)
    return None


def loadwav(name):
    """mel/impl.go:234-264: first channel of a PCM WAV as float64 in [-1, 1); (empty, 0) when it cannot be decoded."""
    try:
        with open(name, "rb") as f:
            data = f.read()
        if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
            raise ValueError("not a RIFF/WAVE file")
        pos, fmt, pcm = 12, None, None
        while pos + 8 <= len(data):
            tag, n = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
            body = data[pos + 8:pos + 8 + n]
            pos += 8 + n + (n & 1)
            if tag == b"fmt ":
                fmt = struct.unpack("<HHIIHH", body[:16])
            elif tag == b"data":
                pcm = body
        if fmt is None or pcm is None or fmt[0] not in (1, 0xFFFE):
            raise ValueError("unsupported WAV")
        _, ch, sr, _, _, bits = fmt
        # scales as beep v1.1.0's wav decoder applies them (restated from memory, source absent): x / (2^bits - 1) * 2
        if bits == 8:
            a = np.frombuffer(pcm, dtype=np.uint8).astype(np.float64) / 255.0 * 2.0 - 1.0
        elif bits == 16:
            a = np.frombuffer(pcm[:len(pcm) // 2 * 2], dtype="<i2").astype(np.float64) / 65535.0 * 2.0
        elif bits == 24:
            b = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            a = ((b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) << 8 >> 8).astype(np.float64) / 16777215.0 * 2.0
        else:
            raise ValueError("unsupported WAV sample size")
        a = a[:len(a) // ch * ch].reshape(-1, ch)[:, 0]
        return np.ascontiguousarray(a), float(sr)
    except (OSError, ValueError, struct.error) as e:
        print(str(e))
        return np.zeros(0), 0.0


def loadflac(name):
    """mel/impl.go:266-296 (github.com/mewkiz/flac in the reference; gomel_amd/flac.py here): every sample of every subframe of every
    frame, scaled by 1/(256*256) whatever the sample size; (empty, 0) with the error printed when the file cannot be decoded."""
    from . import flac
    try:
        return flac.load(name)
    except (OSError, ValueError, IndexError, struct.error) as e:
        print(str(e))
        return np.zeros(0), 0.0

