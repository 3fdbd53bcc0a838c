"""FLAC loader (`loadflac`, mel/impl.go:266-296; the reference calls github.com/mewkiz/flac): the RFC 9639 example file as a
known-answer test (frame CRCs and the STREAMINFO MD5 check themselves), then format coverage through the test-side encoder,
then the reference's own quirks: every subframe of every frame appended in turn, 1/65536 whatever the sample size."""
import hashlib
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flac_writer as fw  # noqa: E402


def test_rfc9639_example_file_known_answer():
    """RFC 9639 Appendix D.1: a complete 57-byte FLAC file holding one stereo 16-bit sample.  Its header CRC-8 (0xbf), frame CRC-16
    (0xaa9a) and the MD5 of the decoded audio stored in STREAMINFO all have to agree with what the decoder computes."""
    from gomel_amd import flac
    data = bytes.fromhex("664c614380000022100010000000" "0f00000f" "0ac442f0" "00000001" "3e84b41807dc690307586a3dad1a2e0f"
                         "fff869180000bf" "0358fd" "03128b" "aa9a")
    info, frames = flac.decode(data)
    assert (info["sample_rate"], info["channels"], info["bits_per_sample"], info["total_samples"]) == (44100, 2, 16, 1)
    assert frames == [[[25588], [10416]]]                                    # both subframes VERBATIM with wasted bits (2 and 4)
    pcm = b"".join(struct.pack("<hh", l, r) for l, r in zip(frames[0][0], frames[0][1]))
    assert hashlib.md5(pcm).digest() == info["md5"]
    bad = bytearray(data)
    bad[-3] ^= 1
    with pytest.raises(flac.FlacError):
        flac.decode(bytes(bad))                                              # a flipped audio bit fails the frame CRC-16


def _sig(n, seed, amp):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    return np.round(amp * (0.6 * np.sin(0.031 * t) + 0.3 * np.sin(0.0071 * t * (1 + seed)) + 0.02 * rng.standard_normal(n))).astype(np.int64).tolist()


@pytest.mark.parametrize("bps", [8, 16, 24])
def test_every_subframe_type_and_channel_assignment(bps):
    from gomel_amd import flac
    amp = (1 << (bps - 1)) * 0.7
    n = 1152
    left, right = _sig(n, 1, amp), _sig(n, 2, amp * 0.8)
    frames, want = [], []
    specs = [
        ("independent", [dict(kind="verbatim"), dict(kind="constant")], [left, [37] * n]),
        ("independent", [dict(kind="fixed", order=0, porder=2), dict(kind="fixed", order=1, porder=0, method=1)], [left, right]),
        ("left_side", [dict(kind="fixed", order=2, porder=3), dict(kind="fixed", order=3, porder=1)], [left, right]),
        ("side_right", [dict(kind="fixed", order=4, porder=2, escape_part=1), dict(kind="lpc", order=2, coefs=[1900, -900], shift=10, prec=12, porder=2)], [left, right]),
        ("mid_side", [dict(kind="lpc", order=8, coefs=[700, -300, 150, -80, 40, -20, 10, -5], shift=9, prec=11, porder=4, method=1),
                      dict(kind="fixed", order=2, porder=0)], [left, right]),
        ("independent", [dict(kind="verbatim", wasted=3), dict(kind="fixed", order=1, porder=1, wasted=1)], [[v & ~7 for v in left], [v & ~1 for v in right]]),
    ]
    for i, (mode, sp, chans) in enumerate(specs):
        frames.append(fw.frame(chans, bps, i, mode, sp))
        want.append(chans)
    tail = [left[:100], right[:100]]                                          # a short last block (8-bit block-size field)
    frames.append(fw.frame(tail, bps, len(specs), "independent", [dict(kind="fixed", order=2, porder=0)] * 2))
    want.append(tail)
    big = [_sig(300, 5, amp), _sig(300, 6, amp)]                              # 16-bit block-size field, two-byte coded frame number
    frames.append(fw.frame(big, bps, 200, "mid_side", [dict(kind="fixed", order=1, porder=2)] * 2))
    want.append(big)
    data = fw.stream(frames, 22050, 2, bps, sum(len(w[0]) for w in want), blocksize=1152)
    info, got = flac.decode(data)
    assert info["sample_rate"] == 22050 and info["channels"] == 2 and info["bits_per_sample"] == bps
    assert got == want


def test_loadflac_follows_the_reference_quirks(tmp_path):
    """mel/impl.go:283-293: for each frame, for each subframe, every sample / (256*256) -- stereo comes out block-interleaved, and
    24-bit audio is NOT rescaled to [-1, 1)."""
    import gomel_amd
    from gomel_amd import codec
    a, b = _sig(512, 3, 20000), _sig(512, 4, 9000)
    f0 = fw.frame([a[:256], b[:256]], 16, 0, "mid_side", [dict(kind="fixed", order=2, porder=1)] * 2)
    f1 = fw.frame([a[256:], b[256:]], 16, 1, "left_side", [dict(kind="fixed", order=1, porder=0)] * 2)
    p = tmp_path / "stereo.flac"
    p.write_bytes(fw.stream([f0, f1], 44100, 2, 16, 512, blocksize=256, extra_metadata=False))
    out, sr = codec.loadflac(str(p))
    want = np.array(a[:256] + b[:256] + a[256:] + b[256:], dtype=np.float64) / 65536.0
    assert sr == 44100.0 and np.array_equal(out, want)
    assert np.array_equal(gomel_amd.LoadFlac(str(p)), want)
    mono = _sig(4096, 9, 3.0e6)
    q = tmp_path / "mono24.flac"
    q.write_bytes(fw.stream([fw.frame([mono], 24, 0, "independent", [dict(kind="fixed", order=2, porder=4)])], 48000, 1, 24, 4096))
    out, sr = codec.loadflac(str(q))
    assert sr == 48000.0 and np.array_equal(out, np.array(mono, dtype=np.float64) / 65536.0) and np.max(np.abs(out)) > 1.0
    r = tmp_path / "broken.flac"
    r.write_bytes(b"fLaC" + b"\x00" * 10)
    out, sr = codec.loadflac(str(r))                                          # the reference prints the error and returns (nil, 0)
    assert len(out) == 0 and sr == 0.0
    assert len(gomel_amd.LoadFlac(str(tmp_path / "missing.flac"))) == 0


def test_leading_id3v2_tag_and_trailing_bytes(tmp_path):
    """Files taggers have touched: an ID3v2 block in front of `fLaC`, an ID3v1 tag / padding behind the last frame.  mewkiz/flac (the
    reference's decoder, mel/impl.go:266-296) reads them; the frames decoded before the trailing bytes are kept."""
    from gomel_amd import codec, flac
    a = _sig(512, 11, 12000)
    body = fw.stream([fw.frame([a[:256]], 16, 0, "independent", [dict(kind="fixed", order=2, porder=1)]),
                      fw.frame([a[256:]], 16, 1, "independent", [dict(kind="fixed", order=1, porder=0)])], 44100, 1, 16, 512, blocksize=256)
    want = np.array(a, dtype=np.float64) / 65536.0
    id3v2 = b"ID3\x04\x00\x00" + bytes([0, 0, 1, 5]) + b"\x00" * 133          # synch-safe size 1 << 7 | 5 = 133
    id3v1 = b"TAG" + b"\x00" * 125
    for name, data in (("tagged.flac", id3v2 + body), ("trailing.flac", body + id3v1), ("both.flac", id3v2 + body + b"\x00" * 4096)):
        p = tmp_path / name
        p.write_bytes(data)
        out, sr = codec.loadflac(str(p))
        assert sr == 44100.0 and np.array_equal(out, want), name
    # no total_samples in STREAMINFO (streamed encodes): the trailing bytes end the decode at the first non-frame
    info, frames = flac.decode(fw.stream([fw.frame([a[:256]], 16, 0, "independent", [dict(kind="fixed", order=2, porder=1)])], 44100, 1, 16, 0, blocksize=256) + id3v1)
    assert info["total_samples"] == 0 and frames == [[a[:256]]]
    with pytest.raises(flac.FlacError):
        flac.decode(b"ID3\x04\x00\x00\x00\x00\x00\x02xxRIFF")                # an ID3 tag in front of something that is not FLAC


@pytest.mark.gpu
def test_to_mel_flac_pipeline(tmp_path, oracle):
    """(*Mel).ToMelFlac, mel/mel.go:176-192: loadflac -> ToMel -> dumpimage; an unreadable file is ErrFileNotLoaded."""
    import gomel_amd
    from gomel_amd import codec
    sig = oracle.config2_signal(sr=22050, seconds=0.6)
    pcm = np.round(sig * 20000).astype(np.int64).tolist()
    frames = [fw.frame([pcm[i:i + 4096]], 16, k, "independent", [dict(kind="fixed", order=2, porder=3 if len(pcm[i:i + 4096]) % 8 == 0 else 0)])
              for k, i in enumerate(range(0, len(pcm), 4096))]
    p = tmp_path / "in.flac"
    p.write_bytes(fw.stream(frames, 22050, 1, 16, len(pcm)))
    m = gomel_amd.NewMel()
    m.NumMels, m.MelFmax, m.Window, m.Resolut = 80, 8000.0, 256, 1024
    out = tmp_path / "out.png"
    m.ToMelFlac(str(p), str(out))
    buf, samples, sr = codec.loadpng(str(out), m.YReverse)
    assert sr == float(np.float16(22050.0)) and len(buf) == len(m.ToMel(np.array(pcm, dtype=np.float64) / 65536.0))   # the PNG metadata is float16 (impl.go:120-125)
    with pytest.raises(gomel_amd.ErrFileNotLoaded):
        m.ToMelFlac(str(tmp_path / "nothing.flac"), str(out))
