"""CPU tests of the host-side wire formats (PNG + float16 metadata, WAV) against the oracle's restatement and the fixtures."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CHIRP = os.path.join(HERE, "golden", "chirp_192.png")
BANNER = os.path.join(HERE, "golden", "reference_banner_183x80.png")   # the reference's README banner (data file)


def test_float16_metadata_roundtrip():
    from gomel_amd import codec
    for v in (0.0, 1.0, -11.5, 44100.0, 22050.0, 3.0637e-05, 65504.0, 1e-8):
        b = codec.packFloat16ToBytes(v)
        assert len(b) == 2
        got = codec.unpackBytesToFloat64(b)
        assert got == float(np.float16(np.float32(v)))
    assert codec.packFloat16ToBytes(1.0) == b"\x00\x3c"                      # little endian 0x3C00
    assert abs(codec.unpackBytesToFloat64(b"\x02\x02") - 3.0637e-05) < 1e-9


def test_loadpng_matches_oracle_on_fixture(oracle):
    from gomel_amd import codec
    buf, samples, sr = codec.loadpng(CHIRP, True)
    img = codec.read_png(CHIRP)
    want, wsamples, wsr = oracle.loadpng_pixels(img, True)
    assert img.shape == (192, 39, 4)
    assert np.array_equal(buf, want) and samples == wsamples and sr == wsr == 44096.0   # float16(44100) = 44096: the reference's own quirk
    assert buf.shape == (39 * 192, 2)
    assert samples > 0 and abs(samples - 52920) / 52920 < 2e-3               # 1.2 s @ 44.1 kHz, float16 precision


def test_dumpimage_pixels_match_oracle_and_roundtrip(oracle, tmp_path):
    from gomel_amd import codec
    rng = np.random.default_rng(0)
    mel = rng.uniform(-11.5, 4.0, size=(13 * 24, 2))
    for rev in (False, True):
        p = str(tmp_path / ("t%d.png" % rev))
        codec.dumpimage(p, mel, 24, rev, 1280.0, 44100.0)
        assert np.array_equal(codec.read_png(p), oracle.dumpimage_pixels(mel, 24, rev, 1280.0, 44100.0))
        back, samples, sr = codec.loadpng(p, rev)
        assert back.shape == mel.shape and sr == 44096.0 and samples == 1280.0 * 13
        step = (float(np.float16(mel.max())) - float(np.float16(mel.min()))) / 255.0
        assert np.max(np.abs(back - mel)) < 1.5 * step + 0.02               # 8-bit truncation + float16 min/max
    buf, s, r = codec.loadpng(str(tmp_path / "missing.png"), False)
    assert len(buf) == 0 and s == 0 and r == 0                               # reference: println(err), return nil


@pytest.mark.skipif(not os.path.exists(BANNER), reason="banner fixture absent")
def test_reference_banner_is_legacy_and_unusable_for_towav(oracle):
    """183x80 RGB, metadata bytes all 2 -> max == min == samples: constant decode, samples 0; and
    183*80 pairs is not a multiple of NumMels = 192 (the reference panics; the C ABI returns GOMEL_ERR_BAD_LENGTH)."""
    from gomel_amd import codec
    import ctypes
    from gomel_amd import _lib
    img = codec.read_png(BANNER)
    assert img.shape == (80, 183, 4)
    buf, samples, sr = codec.loadpng(BANNER, True)
    assert len(buf) == 183 * 80 and samples == 0
    assert np.allclose(buf, buf[0, 0]) and abs(buf[0, 0] - 3.0637e-05) < 1e-8
    c = _lib.GomelConfig()
    _lib.lib().gomel_default_config(ctypes.byref(c))
    c.num_mels = 192
    assert _lib.lib().gomel_out_len(ctypes.byref(c), len(buf)) == 0


def test_wav_roundtrip_and_clamp(oracle, tmp_path):
    from gomel_amd import codec
    x = np.concatenate([np.linspace(-1.5, 1.5, 1001), [0.25, -0.25]])
    p = str(tmp_path / "a.wav")
    codec.dumpwav(p, x, 22050)
    raw = open(p, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt "
    pcm = np.frombuffer(raw[44:], dtype="<i2")
    assert np.array_equal(pcm, oracle.wav_pcm16(x))                          # clamp to [-1,1], * 32767, truncate
    assert pcm.max() == 32767 and pcm.min() == -32767
    y, sr = codec.loadwav(p)
    assert sr == 22050.0 and len(y) == len(x)
    assert np.max(np.abs(y - np.clip(x, -1, 1))) < 1e-4
    buf, sr = codec.loadwav(str(tmp_path / "nope.wav"))
    assert len(buf) == 0 and sr == 0


def test_dumpbuffer_image(oracle):
    from gomel_amd import codec
    rng = np.random.default_rng(1)
    mel = rng.uniform(-5, 2, size=(6 * 8, 2))
    img = codec.dumpbuffer(mel, 8)                                           # (*Mel).Image, mel/mel.go:171-173
    assert img.dtype == np.uint16 and img.shape == (48,)
    lo = (255 * ((mel[:, 0] - mel[:, 0].min()) / (mel[:, 0].max() - mel[:, 0].min()))).astype(int)   # impl.go:36-38 order
    assert np.array_equal(img & 255, lo)
