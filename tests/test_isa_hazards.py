"""CPU test: the gfx950 ISA of the HIP library honours the "VALU that uses OPSEL -> 1 wait state before a consumer" rule
(CDNA ISA, manually inserted wait states) everywhere, including inside and between the hand-written packed-math asm blocks of
gl_fft.hpp, where hipcc's own hazard recogniser does not look.  Also: no scratch (spill) traffic in the fused kernel's loop."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gomel_amd", "csrc", "gomel_hip.hip")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "gomel.s")
    # the product's flags (gomel_amd/build.py), so that what is audited is what ships
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt",
                    "-S", "--cuda-device-only", "-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
    return [l.rstrip() for l in open(out)]


def _regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _instructions(lines):
    for l in lines:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or t.startswith("//"):
            continue
        yield t.split(";")[0].strip()


def test_asm_blocks_never_consume_an_opsel_result_in_the_next_instruction(isa):
    """Conservative form of the rule, applied to the hand-written blocks only (hipcc pads its own code, and pads a block's
    outputs before ITS next instruction, but knows nothing about the inside of a block or about a block that directly
    follows another): within a block, and from the last instruction of a block to the first of a directly following one, no
    instruction may read the destination of an immediately preceding packed op that carries op_sel."""
    blocks, cur, prev_end = [], None, None
    seq = []                                    # (kind, text): kind in {"asm", "code"}
    for l in isa:
        t = l.strip()
        if "ASMSTART" in t:
            cur = []
            continue
        if "ASMEND" in t:
            seq.append(("asm", cur))
            cur = None
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if cur is not None:
            cur.append(t)
        else:
            seq.append(("code", t))
    n_blocks = n_pairs = 0
    bad = []

    def hazard(a, b):
        if not (a.startswith("v_pk_") and "op_sel" in a) or not b.startswith("v_"):
            return False
        return bool(_regs(a.split(",")[0]) & _regs(",".join(b.split(None, 1)[1].split(",")[1:])))

    for i, (kind, item) in enumerate(seq):
        if kind != "asm" or not item:
            continue
        n_blocks += 1
        for a, b in zip(item, item[1:]):
            n_pairs += 1
            if hazard(a, b):
                bad.append((a, b))
        if i + 1 < len(seq) and seq[i + 1][0] == "asm" and seq[i + 1][1]:
            if hazard(item[-1], seq[i + 1][1][0]):
                bad.append((item[-1], seq[i + 1][1][0]))
    assert n_blocks > 300 and n_pairs > 2000      # the packed forms are in use in every instantiation
    assert not bad, bad[:5]


def test_no_fused_kernel_instantiation_spills(isa):
    """k_gl_fused runs at 2 waves per SIMD on 224-256 VGPRs: how many twiddles each geometry keeps in registers is chosen per
    instantiation (fused_w1_keep / fused_w2_regs / fused_late_rows) so that NONE of the 25 spills; a spill would put scratch traffic
    into the frame loop.  The kernel metadata of the compiled code object says so directly."""
    text = "\n".join(isa)
    found = re.findall(r"\.name:\s+(_Z10k_gl_fusedILi(\d+)ELi(\d+)E\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(found) == 25, len(found)
    for name, n, sh, vgpr, spill in found:
        assert int(spill) == 0 and int(vgpr) <= 256, (n, sh, vgpr, spill)
    body = isa[next(i for i, l in enumerate(isa) if l.startswith("_Z10k_gl_fusedILi4096ELi5EE") and ":" in l):]
    body = body[:next(i for i, l in enumerate(body) if "s_endpgm" in l)]
    assert not any("scratch_" in l for l in body)
    assert sum("global_load_dwordx4" in l for l in body) >= 6      # four magnitude loads + two row pairs per frame


def test_persistent_kernel_builds_fit_their_occupancy(isa):
    """k_gl_persist must have every stream resident: the OCC 1 build may use the whole 512-VGPR file of a lone wave, the OCC 2 build
    must stay within 256, and neither may spill (the host's occupancy census, and with it the path choice, rests on these numbers).
    Its exchange must be what DESIGN.md says: write-through (sc1) 16-byte stores, sc1 16-byte loads, no fence."""
    text = "\n".join(isa)
    found = re.findall(r"\.name:\s+(_Z12k_gl_persistILi(\d+)ELi(\d+)ELi(\d+)E\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(found) == 10, [f[0] for f in found]          # five frame lengths x (one wave per SIMD, two)
    for name, n, occ, depth, vgpr, spill in found:
        assert (int(occ), int(depth)) in ((1, 4), (2, 2))
        assert int(spill) == 0 and int(vgpr) <= 512 // int(occ), (n, occ, vgpr, spill)
    body = isa[next(i for i, l in enumerate(isa) if l.startswith("_Z12k_gl_persistILi4096ELi1ELi4EE") and ":" in l):]
    body = body[:next(i for i, l in enumerate(body) if "s_endpgm" in l)]
    assert not any("scratch_" in l for l in body)
    stores = [l for l in body if "buffer_store_dwordx4" in l]
    loads = [l for l in body if "buffer_load_dwordx4" in l]
    assert len(stores) >= 16 and all(" sc1" in l for l in stores)        # eight sentinel quads + eight fresh quads per iteration
    assert len(loads) >= 24 and all(" sc1" in l for l in loads)          # the gather: up to 32 quads, every one of them write-through-coherent
    assert not any("buffer_inv" in l or "buffer_wbl2" in l for l in body)


def test_any_hop_fused_kernel_has_no_flat_loads_no_spills_no_waits_at_the_loop_top(isa):
    """k_gl_fused_any lost 30 % to three code-generation accidents, each visible in the ISA: a select between an LDS
    and a global ADDRESS became a flat load (after which every LDS read waits for every global load), loads behind per-row branches
    got vmcnt(0) at every join, and the prologue's loads were counted down in front of every frame's LDS writes.  Keep them out."""
    text = "\n".join(isa)
    found = re.findall(r"\.name:\s+(_Z14k_gl_fused_anyILi(\d+)ELi(\d+)ELb[01]E\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(found) == 30, [f[0] for f in found]                     # five frame lengths x three new-row counts x even / odd hop
    for name, n, nr, vgpr, spill in found:
        # (the 16-row builds -- hops above 7N/16, two covering frames -- hold 16 loaded rows beside 16 LDS rows, the odd-hop builds
        # select per float: a few spills at N 4096 there; the builds the usual hops take -- 4 rows, and 8 rows at even hops -- have none)
        odd = "Lb1E" in name
        assert int(nr) in (4, 8, 16) and int(vgpr) <= 256 and int(spill) <= (0 if int(nr) == 4 or (int(nr) == 8 and not odd) else 48), (n, nr, odd, vgpr, spill)
    for inst in ("_Z14k_gl_fused_anyILi4096ELi4ELb0EE", "_Z14k_gl_fused_anyILi2048ELi4ELb0EE", "_Z14k_gl_fused_anyILi1024ELi8ELb0EE", "_Z14k_gl_fused_anyILi4096ELi4ELb1EE",
                 "_Z14k_gl_fused_anyILi2048ELi4ELb1EE"):
        body = isa[next(i for i, l in enumerate(isa) if l.startswith(inst) and ":" in l):]
        body = body[:next(i for i, l in enumerate(body) if "s_endpgm" in l)]
        assert not any("flat_load" in l or "flat_store" in l or "scratch_" in l for l in body), inst
        loop = body[max(i for i, l in enumerate(body) if "Loop Header" in l):]             # the frame loop (the table staging loops come first)
        first_lds_write = next(i for i, l in enumerate(loop) if "ds_write_b64" in l)
        assert not any("s_waitcnt vmcnt" in l for l in loop[:first_lds_write]), inst      # nothing to wait for before the window goes to LDS
        assert sum("s_waitcnt vmcnt(0)" in l for l in loop) <= 2, inst                      # (one where the next window's loads are taken in)


def test_fp64_fused_kernel_instruction_and_synchronisation_budget(isa):
    """k_gl_fused_d<4096,5> is the dominant kernel of a default call and sits on an in-order issue floor: what moves its time
    is the NUMBER of instructions a frame issues.  Pinned here, from the compiled code's frame loop (rare branches excluded): the fp64
    arithmetic of round 5 (1 487 per frame and wave: third-order phasor step, class-based range test, sqrt(1/2) twiddles on the radix-4
    multiply-adds; 1 613 before), the FOUR polls a frame keeps of the eight it had (paired FFTs, gl_fft.hpp PAIRED bits 0-1: six; the in-place
    pass-2 exchange of round 6, bit 2: four -- only "the partner has written" is ever waited for, and nothing is published behind a read burst; a
    poll's bound sits out of line behind a branch and does not count), sixteen v_rsq_f64 and sixteen v_cmp_class_f64, 64 + 149 LDS data accesses of
    16 bytes, no scratch and no flat access (a flat store pending behind a poll's give-up branch turned every counted LDS wait into lgkmcnt(0)), counted
    LDS waits behind the read bursts, one wave per SIMD within 512 registers."""
    text = "\n".join(isa)
    m = re.search(r"\.name:\s+(_Z12k_gl_fused_dILi4096ELi5E\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert m and int(m.group(3)) == 0 and int(m.group(2)) <= 512
    start = next(i for i, l in enumerate(isa) if l.startswith("_Z12k_gl_fused_dILi4096ELi5EE") and ":" in l)
    k = [l.strip() for l in isa[start:start + next(i for i, l in enumerate(isa[start:]) if "s_endpgm" in l) + 1]]
    hdr = next(i for i, l in enumerate(k) if "This Loop Header: Depth=1" in l)
    # the loop's blocks in front of its header (the latch: counter update, the frame's first loads) start at the first label marked as part of the loop;
    # the first branch back to it is the common back edge -- what follows are the last frame's tail stores and the rare branches' blocks
    latch = min(i for i, l in enumerate(k[:hdr]) if re.match(r"^\.LBB\d+_\d+:", l) and "in Loop: Header=" in l)
    label = k[latch].split(":")[0]
    end = next(i for i in range(hdr, len(k)) if re.match(r"^s_c?branch\S* " + re.escape(label) + r"$", k[i]))
    body = list(_instructions(k[latch:end + 1]))
    ops = [b.split()[0] for b in body]
    f64 = sum("f64" in o and not o.startswith("v_rsq") and not o.startswith("v_cmp") for o in ops)
    assert 1400 <= f64 <= 1500, f64
    assert ops.count("v_rsq_f64_e32") == 16 and sum(o.startswith("v_cmp_class_f64") for o in ops) == 16
    assert sum(o == "ds_write_b128" for o in ops) == 64 and 140 <= sum(o == "ds_read_b128" for o in ops) <= 150
    assert sum(o == "ds_read_b32" for o in ops) == 4 and sum(o == "ds_write_b32" for o in ops) == 4      # event counters: four waits, four arrives per frame
    waits = [b for b in body if b.startswith("s_waitcnt") and "lgkmcnt" in b]
    assert sum("lgkmcnt(0)" not in w for w in waits) >= 20, len(waits)                                   # the read bursts stream into the arithmetic (counted waits)
    assert not any(o.startswith("scratch_") or o.startswith("flat_") for o in ops)
    # branches of a frame: the eight range tests of the spectral step cost ONE each (clamp_pair_d's patch form, gl_fft.hpp GL_D_CLAMP_PATCH; the if / else form: two and a flag)
    assert sum(o.startswith("s_cbranch") or o == "s_branch" for o in ops) <= 24, sum(o.startswith("s_cbranch") or o == "s_branch" for o in ops)
    assert sum(o.startswith("v_accvgpr") for o in ops) <= 175 and sum(o.startswith("v_mov") for o in ops) <= 110   # parking / slide traffic: keep an eye on it
