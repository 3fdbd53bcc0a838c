"""The exchange-layout facts the paired FFT forms rely on (gl_fft.hpp, `PAIRED`; rounds 5 and 6), checked by plain arithmetic on the geometry
constants read from the source: between a forward FFT's last reads and the following inverse's first writes, and between an inverse's
last reads and the next forward's first writes, NO synchronisation is needed only because each slot involved is read and then written by
the SAME lane and by no other.  Replaces, per frame, go-dsp's fft.FFTReal / fft.IFFT as the reference calls them (mel/mel.go:95,116)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "gomel_amd", "csrc", "gl_fft.hpp")).read()


def geometries():
    """{N: (R1, R2)} from the Geo<N> specialisations; the stride formulas from GeoBase (asserted to be what this test assumes)."""
    geo = {int(n): (int(r1), int(r2)) for n, r1, r2 in re.findall(r"template <> struct Geo<(\d+)> : GeoBase<\d+, (\d+), (\d+)> \{\};", SRC)}
    assert sorted(geo) == [256, 512, 1024, 2048, 4096]
    for formula in ("SA = C1 + 32 / R1", "SAI = C1 + 16 / R1", "SB = B3 + R1", "B3 = C / R3", "T = C / P"):
        assert formula in SRC, formula
    return geo


def test_pass3_slots_have_exactly_one_owner():
    """Forward pass 3 reads lds[r * SB + bA], lds[r * SB + bB] (bA = t, bB = t ? B3 - t : B3 / 2) and the inverse's first pass writes the
    same expressions: over all lanes t and rows r the slots are pairwise distinct, so a lane only ever overwrites what it alone has read."""
    for N, (R1, R2) in geometries().items():
        C = N // 2; T = C // 16; B3 = C // 8; SB = B3 + R1
        seen = {}
        for t in range(T):
            bA, bB = t, (B3 - t if t else B3 // 2)
            for r in range(8):
                for slot in (r * SB + bA, r * SB + bB):
                    assert slot not in seen, (N, t, r, seen[slot])
                    seen[slot] = t
        assert len(seen) == C                                    # every point of the frame, once


def test_layout_a_with_one_stride_is_lane_private_at_the_frame_boundary():
    """PAIRED bit 1: with ONE row stride (SAI) for layout A in both directions the inverse's last reads (lds[q * SAI + t1], t1 = t + T u)
    and the next forward's first writes (the same expression) touch the same slots from the same lane, all slots distinct; and the stride
    keeps every slot inside the stream's exchange area."""
    for N, (R1, R2) in geometries().items():
        C = N // 2; T = C // 16; C1 = C // R1; NB1 = 16 // R1; B3 = C // 8
        SA, SAI, SB = C1 + 32 // R1, C1 + 16 // R1, B3 + R1
        lds_slots = max(R1 * SA, 8 * SB)
        owner = {}
        for t in range(T):
            for u in range(NB1):
                t1 = t + T * u
                assert t1 < C1 <= SAI
                for q in range(R1):
                    slot = q * SAI + t1
                    assert slot < lds_slots and slot not in owner, (N, t, q)
                    owner[slot] = t
        assert len(owner) == C


def test_kernels_that_keep_their_exchange_area_to_themselves_use_the_paired_forms():
    """The product builds pair the FFTs where a frame (or iteration) loop touches the exchange area through these two calls alone: the fused
    kernels (fp64 at both joints, fp32 at the pass-3 joint: profiles/r05/ab_fp64_round5_knobs.txt) and the persistent kernels (same choice per
    precision; their gather and stores go through global memory).  The any-hop kernels shift rows through the area between frames and k_gl_frames
    runs one forward / inverse pair per launch: they take the pass-3 joint only; ToMel / phase / the write-back run a single FFT."""
    k = open(os.path.join(ROOT, "gomel_amd", "csrc", "gl_kernels.hpp")).read()
    assert "#define GL_PAIRED 1" in k and "#define GL_D_PAIRED 3" in k and "#define GL_D_INPLACE 1" in k and "#define GL_INPLACE 0" in k
    # (round 6: the bits come from d_paired<G>() / f_paired<G>() -- GL_D_PAIRED resp. GL_PAIRED plus bit 2, the in-place exchange, where the
    # geometry has one butterfly per lane and pass and the knob is on: fp64 yes, fp32 no)
    assert k.count("fft_forward<G, d_paired<G>()>") == 3 and k.count("fft_inverse<G, d_paired<G>()>") == 2      # k_gl_fused_d (two twiddle forms of the forward call), k_gl_persist_d
    assert k.count("fft_forward<G, f_paired<G>()>") == 2 and k.count("fft_inverse<G, f_paired<G>()>") == 2      # k_gl_fused, k_gl_persist
    assert k.count("fft_forward<G, (GL_PAIRED & 1)>") == 2 and k.count("fft_inverse<G, (GL_PAIRED & 1)>") == 2   # pass-3 joint only: k_gl_frames, k_gl_fused_any
    assert k.count("fft_forward<G, g_paired_d<G>()>") == 2 and k.count("fft_inverse<G, g_paired_d<G>()>") == 2   # ... and their fp64 forms (bit 2 measured there: slower, GL_D_INPLACE_ANY 0)
    assert "#define GL_D_INPLACE_ANY 0" in k
    assert len(re.findall(r"fft_(forward|inverse)<G, [^>]+>", k)) == 17


def test_in_place_exchange_gives_every_slot_one_owner_in_every_pass():
    """PAIRED bit 2 (round 6, N 4096: one butterfly per lane and pass): pass 1 writes / the inverse's pass 1 reads slot q * SAI + t; pass 2 reads AND
    writes q1 * SAI + t2 + 8 m with (q1, t2) = (t % R1, t / R1); pass 3 reads / the inverse's pass 3 writes (b % R1) * SAI + 8 (b / R1) + r for the
    lane's two mirrored butterflies b.  In each pass the lanes' slot sets are pairwise disjoint and together are the same C slots: a lane only ever
    overwrites what it alone reads in that pass, which is why the frame needs no "the partner has read" synchronisation.  And the element each pass
    finds in a slot is the one the previous pass put there: pass 2's output q2 of lane (q1, t2) is what butterfly b = q1 + R1 q2 takes as operand t2."""
    assert "lds[inpl_slot<G>(q1, t2 + 8 * q)]" in SRC and "inpl_slot<G>(bA % G::R1, 8 * (bA / G::R1) + r)" in SRC and "return q * S + (GL_D_INPL_XOR" in SRC
    N = 4096
    R1, R2 = geometries()[N]
    C = N // 2; T = C // 16; C1 = C // R1; B3 = C // 8; SAI = C1 + 16 // R1
    assert (R1, R2, 16 // R1, 16 // R2) == (16, 16, 1, 1)
    slot = lambda q, c: q * SAI + c
    p1 = {t: {slot(q, t) for q in range(R1)} for t in range(T)}
    p2 = {t: {slot(t % R1, t // R1 + 8 * m) for m in range(R2)} for t in range(T)}
    p3 = {}
    for t in range(T):
        bs = (t, B3 - t if t else B3 // 2)
        p3[t] = {slot(b % R1, 8 * (b // R1) + r) for b in bs for r in range(8)}
    allslots = set().union(*p1.values())
    assert len(allslots) == C and max(allslots) < max(R1 * (C1 + 32 // R1), 8 * (B3 + R1))     # inside the stream's exchange area (LDS_SLOTS)
    for sets in (p1, p2, p3):
        assert all(len(v) == 16 for v in sets.values())
        assert sum(len(v) for v in sets.values()) == C and set().union(*sets.values()) == allslots     # a partition of the same C slots
    # contents: element (q1, t1) after pass 1 sits at slot(q1, t1); pass 2's lane (q1, t2) takes t1 = t2 + 8 m and leaves (q1, t2, q2) at slot(q1, t2 + 8 q2);
    # butterfly b = q1 + R1 q2 of pass 3 takes its operand t2 = r from slot(q1, 8 q2 + r) = slot(b % R1, 8 (b // R1) + r)
    for q1 in range(R1):
        for t2 in range(8):
            for q2 in range(R2):
                b = q1 + R1 * q2
                assert slot(q1, t2 + 8 * q2) == slot(b % R1, 8 * (b // R1) + t2)
