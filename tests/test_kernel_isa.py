"""CPU: a build-time look at the ISA hipcc emits for the hand-scheduled kernels.

Their speed rests on properties the source cannot express and a compiler upgrade (or an innocent
edit) can silently break: NO scratch traffic inside the steady-state loop (a spilled fragment is
reloaded behind an `s_waitcnt vmcnt(0)` that also waits for every LDS-DMA piece in flight), the
accumulators staying in the accumulator file (no per-step v_accvgpr shuttles), and the LDS
fragment-read count per MFMA that each design claims.  The kernel source is compiled to assembly
(device only, seconds) and the largest loop of each kernel is inspected."""
import os
import re
import subprocess
from collections import Counter

import pytest

from conftest import PKG, ROOT

HIPCC = "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, PKG, "csrc")


def device_asm(tmp_path_factory, src):
    out = str(tmp_path_factory.mktemp("isa") / (src + ".s"))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fvisibility=hidden",
                           "-Wno-unused-result", "-Wno-inline-asm", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out])
    return open(out).read().split("\n")


def kernel_lines(lines, name_re):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4sdpa\S*" + name_re + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def main_loop_mix(k):
    return main_loop_span(k)[2]


def main_loop_span(k):
    """(first line, last line, instruction histogram) of the INNERMOST backward-branch loop of a kernel that holds the most MFMAs
    (a loop that contains another backward branch is an outer loop: the fp32 kernels' second pass
    around the whole K/V walk, for instance)"""
    labels = {m.group(1): i for i, l in enumerate(k) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    back = []                                          # (target line, branch line) of every backward branch
    for i, l in enumerate(k):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            back.append((labels[m.group(1)], i))

    def mix(lo, hi):
        c = Counter()
        for l in k[lo:hi + 1]:
            m = re.match(r"^\t([a-z_0-9]+)", l)
            if m:
                c[m.group(1)] += 1
        return c

    def mfmas(lo, hi):
        return sum(1 for l in k[lo:hi + 1] if l.startswith("\tv_mfma"))

    best = None
    for lo, hi in back:
        # (a backward branch without MFMAs inside is block layout, not a loop of interest: hipcc places the
        #  cold ragged-tile DMA block of the stream-K instantiations behind its join block)
        if any((lo2, hi2) != (lo, hi) and lo <= lo2 and hi2 <= hi and mfmas(lo2, hi2) > 0 for lo2, hi2 in back):
            continue                                   # contains another loop
        if any(l.startswith("\ts_endpgm") for l in k[lo:hi + 1]):
            continue                                   # block layout again: a tail placed behind the epilogue jumps back
        c = mix(lo, hi)
        n = sum(v for name, v in c.items() if name.startswith("v_mfma"))
        # ties go to the FIRST loop in program order: the fp32 pipelined kernels carry their K/V walk
        # twice (straight-line), and the second copy is the rare eager-rescale pass behind a failed
        # range check -- it may spill, the first one may not
        if best is None or n > best[0]:
            best = (n, (lo, hi), c)
    assert best is not None, "no loop found"
    return best[1][0], best[1][1], best[2]


@pytest.fixture(scope="module")
def bf16_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc here")
    return device_asm(tmp_path_factory, "sdpa_fwd_bf16.hip")


@pytest.mark.parametrize("dk,dv", [(64, 64), (64, 128), (64, 256), (128, 64), (128, 128), (128, 256),
                                   (256, 64), (256, 128), (256, 256)])
def test_duo_kernel_loop_is_spill_free_and_reads_one_fragment_per_two_mfmas(dk, dv, bf16_asm):
    c = main_loop_mix(kernel_lines(bf16_asm, "fused_bf16_duo_kernelILi%dELi%dE" % (dk, dv)))
    mfma = c["v_mfma_f32_32x32x16_bf16"]
    assert mfma == 2 * (2 * dk // 16 + 4 * dv // 32), c          # the loop body is two steps
    assert sum(v for k, v in c.items() if k.startswith("scratch_")) == 0, "scratch traffic inside the loop: %s" % c
    assert c["ds_read_b128"] * 2 == mfma, "every LDS fragment must feed two MFMAs: %s" % c
    # at most one O tile (16 registers) crosses the back-edge through VGPRs (dv = 256: the file is full)
    assert c["v_accvgpr_read_b32"] <= 16 and c["v_accvgpr_write_b32"] <= 16, c
    # softmax VALU of two steps (VALU issue does not overlap MFMA issue on a SIMD, so the count is the
    # cost): 64 exp2, 64 row-sum adds, 32 bf16 packs, and no row max
    assert c["v_exp_f32"] == 64 and c["v_add_f32"] == 64 and c["v_cvt_pk_bf16_f32"] == 32, c
    assert c["v_max3_f32"] == 0 and c["v_max_f32_e32"] == 0 and c["v_pk_add_f32"] == 0, c
    assert c["global_load_lds_dwordx4"] == 2 * (dk // 64 + dv // 64), c      # DMA pieces of two steps


def _loop_ops(k, lo, hi):
    """(mnemonic, operand text, inside an asm block?) of every instruction of k[lo..hi]"""
    out, inside = [], False
    for l in k[lo:hi + 1]:
        if "#ASMSTART" in l:
            inside = True
        elif "#ASMEND" in l:
            inside = False
        elif (m := re.match(r"^\t([a-z_0-9]+)\s*(.*)", l)):
            out.append((m.group(1), m.group(2).split(";")[0], inside))
    return out


def _regs(text):
    """every vector register a token list names: v5, v[0:15] -> {5}, {0..15}"""
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def test_tandem_kernel_loop_is_spill_free_and_reads_50_fragments_per_64_mfmas(bf16_asm):
    """dv > 256: two waves share 64 rows and split the value columns.  Per step and wave: 32 + 32 MFMAs, 32 K + 16 Vt + 2 P
    fragment reads, 2 P stores, 16 DMA pieces; O stays in the accumulator file.  Round 6 (tiled images): ONE barrier per step,
    and the DMA costs one VMEM instruction per piece plus a handful of scalar ones per TILE (<= 1 SALU per piece)."""
    k = kernel_lines(bf16_asm, "fused_bf16_tandem_kernelILi512E")
    lo, hi, c = main_loop_span(k)
    assert c["v_mfma_f32_32x32x16_bf16"] == 128 and c["ds_read_b128"] == 100 and c["ds_write_b128"] == 4, c
    assert sum(v for name, v in c.items() if name.startswith("scratch_")) == 0, c
    # NO accumulator tile crosses the back edge through architectural VGPRs (pin_o() at every step + the block split behind
    # the barrier: 16 + 16 v_accvgpr moves and an MFMA drain per two steps otherwise)
    assert c["v_accvgpr_read_b32"] == 0 and c["v_accvgpr_write_b32"] == 0, c
    assert c["global_load_lds_dwordx4"] == 32 and c["s_barrier"] == 2 and c["v_exp_f32"] == 32, c
    # no ragged-tile mask, no row clamp, no per-piece swizzle in the steady state
    assert c["v_cndmask_b32_e32"] + c["v_cndmask_b32_e64"] == 0 and c["s_min_i32"] == 0 and c["v_xor_b32"] + c["v_xor_b32_e32"] == 0, c
    # scalar work of the DMA: M0 once per tile (4 per two steps), the tile bases, the loop's own counter and branches
    salu = sum(v for name, v in c.items() if name.startswith("s_") and name not in ("s_waitcnt", "s_nop", "s_barrier"))
    assert c["s_mov_b32"] <= 6 and salu <= 32, (salu, c)          # <= 1 scalar instruction per DMA piece, everything included
    assert c["s_nop"] <= 8, c                                        # (no wait states in front of the chain's links)
    assert sum(c.values()) <= 490, (sum(c.values()), c)             # round 5: 651


def test_tandem_kernel_dma_pieces_are_lane_linear_immediates(bf16_asm):
    """every LDS-DMA piece of the steady state is `global_load_lds_dwordx4 vL, s[b:b+1] offset:IMM` with ONE vector register for all
    of them and immediates -4096 .. 3072: no per-piece address arithmetic is left for the compiler to scatter between the MFMAs"""
    k = kernel_lines(bf16_asm, "fused_bf16_tandem_kernelILi512E")
    lo, hi, _ = main_loop_span(k)
    pieces = [l.strip() for l in k[lo:hi + 1] if l.startswith("\tglobal_load_lds_dwordx4")]
    assert len(pieces) == 32
    vregs = {re.match(r"global_load_lds_dwordx4 (v\d+),", p).group(1) for p in pieces}
    assert len(vregs) == 1, vregs
    offs = sorted(int(re.search(r"offset:(-?\d+)", p).group(1)) if "offset:" in p else 0 for p in pieces)
    assert offs == sorted(list(range(-4096, 4096, 1024)) * 4), offs
    assert len({re.search(r"(s\[\d+:\d+\])", p).group(1) for p in pieces}) <= 4        # K and Vt base of each of the two steps


def test_tandem_kernel_chain_links_need_no_wait_states(bf16_asm):
    """The steady-state chain's MFMAs are asm statements WITHOUT the two leading wait states the other asm MFMAs carry (hipcc's
    hazard recogniser does not see into an asm statement).  That is safe while nothing but an LDS read (covered by s_waitcnt) or
    the previous link writes their operands: no VALU instruction that writes a register a link reads may sit within the two
    instructions in front of it ("VALU write VGPR -> MFMA read": 2 wait states on gfx90a+)."""
    k = kernel_lines(bf16_asm, "fused_bf16_tandem_kernelILi512E")
    lo, hi, _ = main_loop_span(k)
    ops = _loop_ops(k, lo, hi)
    links = [i for i, (op, args, inside) in enumerate(ops) if inside and op == "v_mfma_f32_32x32x16_bf16"]
    assert len(links) == 64, len(links)
    for i in links:
        reads = _regs(ops[i][1])
        for op, args, _ in ops[max(0, i - 2):i]:
            if op.startswith("v_") and not op.startswith("v_mfma"):
                dst = args.split(",")[0]
                assert not (_regs(dst) & reads), "VALU write of a link's operand right in front of it: %s %s" % (op, args)


def test_tandem_stream_kernel_loop_does_the_classic_loops_work(bf16_asm):
    """The persistent form (round 5: the same text from sdpa_fwd_bf16_tandem.inc with waits for ready words) must pay for its waits
    with scalar instructions only: per two steps the same 128 MFMAs, 100 fragment reads, 4 P stores, 32 DMA pieces, 4 barriers, 32
    exponentials as the classic loop, no scratch, no accumulator tile through VGPRs -- the polling (a system-scope load, a sleep,
    the acquire's invalidate) sits on a branch the steady state does not take."""
    c = main_loop_mix(kernel_lines(bf16_asm, "fused_bf16_tandem_stream_kernelILi512E"))
    classic = main_loop_mix(kernel_lines(bf16_asm, "fused_bf16_tandem_kernelILi512E"))
    assert c["s_barrier"] == 2, c
    for k in ("ds_write_b128", "global_load_lds_dwordx4", "s_barrier", "v_exp_f32"):
        assert c[k] == classic[k], (k, c[k], classic[k])
    # (hipcc may lay the loop out ROTATED: the XB = 4 P.V MFMAs behind the barrier and the three K fragment reads in front of them
    #  then sit in the latch block outside the span this test measures -- same work per trip)
    assert classic["v_mfma_f32_32x32x16_bf16"] - 4 <= c["v_mfma_f32_32x32x16_bf16"] <= classic["v_mfma_f32_32x32x16_bf16"], c
    assert classic["ds_read_b128"] - 3 <= c["ds_read_b128"] <= classic["ds_read_b128"], c
    assert sum(v for k, v in c.items() if k.startswith("scratch_")) == 0, c
    assert c["v_accvgpr_read_b32"] == 0 and c["v_accvgpr_write_b32"] == 0, c
    assert c["buffer_inv"] <= 2 and c["global_load_dword"] <= 4, c            # (the polls: once in the loop's text)
    assert sum(c.values()) <= sum(classic.values()) + 90, (sum(c.values()), sum(classic.values()))


@pytest.fixture(scope="module")
def f32_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc here")
    return device_asm(tmp_path_factory, "sdpa_fwd_f32.hip")


def test_f32_pipelined_kernel_loop_is_spill_free(f32_asm):
    c = main_loop_mix(kernel_lines(f32_asm, "fused_pipelined_kernelILi128ELi128ELi0ELi0EE"))
    assert c["v_mfma_f32_32x32x2_f32"] == 256, c                  # two tiles x (64 + 64) MFMAs per wave
    assert sum(v for k, v in c.items() if k.startswith("scratch_")) == 0, c
    assert c["v_exp_f32"] == 32 and c["global_load_lds_dwordx4"] == 32, c          # 2 tiles x (8 K + 8 V) 1-KiB pieces


# Round 5: the hot loop of the metric shape's kernel is pinned to the ALLOCATION rounds 1-3 measured 0.914-0.916 with.
# Round 4 made stream-K a template parameter of the same kernel; the classic instantiation then compiled from a
# different token stream, hipcc numbered its registers differently (same instruction mix: the tests above stayed
# green) and every round-4 measurement of the metric shape was 0.8-1.4 % slower.  The body now lives in
# sdpa_fwd_f32_pipelined.inc, included once per form, and the classic form's text is round 3's again.  The
# fingerprint is over the loop's instructions WITH their registers; it is tied to the compiler it was taken with
# (another hipcc allocates differently by right: re-measure, then re-pin).
PINNED_HIPCC = "7.2.26015"
PINNED_F32_LOOP = {"fused_pipelined_kernelILi128ELi128ELi0ELi0EE": ("4f2feff63055857c", 917)}


def test_f32_metric_shape_hot_loop_is_the_round3_register_allocation(f32_asm):
    import hashlib
    ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    if PINNED_HIPCC not in ver:
        pytest.skip("fingerprint was taken with hipcc %s" % PINNED_HIPCC)
    for name, (want, count) in PINNED_F32_LOOP.items():
        k = kernel_lines(f32_asm, name)
        lo, hi, _ = main_loop_span(k)
        body = [re.sub(r"\.LBB\d+_", ".LBB_", l.split(";")[0].strip()) for l in k[lo:hi + 1]]
        body = [l for l in body if l and not l.startswith(".")]
        got = hashlib.sha256("\n".join(body).encode()).hexdigest()[:16]
        assert (got, len(body)) == (want, count), ("the classic fp32 kernel's hot loop changed (%s, %d instructions): an edit to "
                                                   "sdpa_fwd_f32_pipelined.inc reached the SDPA_PK_SK 0 text; same-box A/B before "
                                                   "re-pinning (tools/gpu_lib_ab.py)" % (got, len(body)))


@pytest.mark.parametrize("dk,dv", [(256, 256), (256, 128), (128, 256)])
def test_f32_pipelined_kernel_one_wave_per_simd_keeps_o_in_the_accumulator_file(dk, dv, f32_asm):
    """dense 256-wide dims: 512 registers per wave, score chains as inline-asm MFMAs with VGPR C/D, O^T
    in AGPRs.  No scratch in the loop, and the only accumulator-file moves are the ones written by hand
    in the (cold) deferred-rescale branch: one read and one write per O^T register and step."""
    c = main_loop_mix(kernel_lines(f32_asm, "fused_pipelined_kernelILi%dELi%dELi0ELi0EE" % (dk, dv)))
    assert c["v_mfma_f32_32x32x2_f32"] == 2 * (dk // 2 + 16 * dv // 32), c       # two tiles per loop body
    assert sum(v for k, v in c.items() if k.startswith("scratch_")) == 0, c
    assert c["v_accvgpr_read_b32"] == dv and c["v_accvgpr_write_b32"] == dv, c  # 2 steps x (dv/32 tiles x 16) / ... cold branch only
    assert c["v_exp_f32"] <= 34 and c["global_load_lds_dwordx4"] == 2 * (dk // 16 + dv // 16), c



@pytest.fixture(scope="module")
def dksplit_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc here")
    return device_asm(tmp_path_factory, "sdpa_fwd_f32_dksplit.hip")


@pytest.mark.parametrize("dks,dvs,qb", [(128, 128, 2), (96, 128, 2), (64, 64, 2), (256, 128, 1), (192, 64, 1), (128, 32, 2),
                                        (96, 96, 2), (192, 192, 1), (256, 256, 1)])      # the dv slices matched to dv = 384 / 768 / 1024
def test_f32_dksplit_pipelined_kernel_keeps_the_softmax_between_the_pv_mfmas(dks, dvs, qb, dksplit_asm):
    """fused_dksplit_pipe_kernel: per 32-key tile and wave DKS/2 x QB score MFMAs + 16 x DVS/32 x QB P.V MFMAs, nothing spilled, one
    barrier -- and the exchange reads, the row max and the exponentials of the NEXT tile sit between the P.V
    MFMAs of this one (one wave per SIMD: only instructions written between two MFMAs run in an MFMA's
    shadow).  Left to itself LLVM sinks that pure VALU work to the loop end, next to its first use."""
    k = kernel_lines(dksplit_asm, "fused_dksplit_pipe_kernelILi%dELi%dELi%dE" % (dks, dvs, qb))
    lo, hi, c = main_loop_span(k)
    assert c["v_mfma_f32_32x32x2_f32"] == qb * (dks // 2 + 16 * dvs // 32), c
    assert sum(v for name, v in c.items() if name.startswith("scratch_")) == 0, c
    assert c["s_barrier"] == 1 and c["ds_write_b128"] == 4 * qb and c["ds_read_b128"] == 16 * qb, c
    ops = [m.group(1) for l in k[lo:hi + 1] if (m := re.match(r"^\t([a-z_0-9]+)", l))]
    mfma_at = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
    inside = lambda name: sum(1 for i, o in enumerate(ops) if o == name and mfma_at[0] < i < mfma_at[-1])
    assert inside("v_exp_f32") >= 17 * qb, c           # 16 exponentials + alpha per block, all before the last MFMA
    assert inside("ds_read_b128") == 16 * qb, c
    # and never more than a few of them in one gap between two MFMAs
    gaps = [sum(1 for o in ops[a + 1:b] if o == "v_exp_f32") for a, b in zip(mfma_at, mfma_at[1:])]
    slots = 14 * qb * dvs // 32      # P.V MFMAs of steps 2..15 (step 0 carries the exchange stores, the barrier sits behind step 1)
    assert max(gaps) <= -(-50 * qb // slots), gaps       # 50 units per query block to place
    # the partial scores go to the exchange buffer between the first P.V MFMAs, not in front of them
    writes = [i for i, o in enumerate(ops) if o == "ds_write_b128"]
    mps = qb * dvs // 32                                # MFMAs of P.V step 0; ceil(4 qb / mps) stores behind each
    per_gap = -(-4 * qb // mps)
    assert sum(1 for a, b in zip(mfma_at, mfma_at[1:]) if any(a < w < b for w in writes)) >= -(-4 * qb // per_gap), writes


@pytest.mark.parametrize("dk,dv", [(64, 64), (128, 64), (64, 128)])
def test_f32_pipelined_kernel_small_dims_first_pass_is_spill_free(dk, dv, f32_asm):
    c = main_loop_mix(kernel_lines(f32_asm, "fused_pipelined_kernelILi%dELi%dELi0ELi0EE" % (dk, dv)))
    assert c["v_mfma_f32_32x32x2_f32"] == 2 * (dk // 2 + 16 * dv // 32), c
    assert sum(v for k, v in c.items() if k.startswith("scratch_")) == 0, c


@pytest.mark.parametrize("dk,dv", [(128, 128), (64, 64), (128, 64), (64, 128), (128, 256)])
def test_f32_stream_k_instantiation_keeps_the_classic_hot_loop(dk, dv, f32_asm):
    """Round 4: fused_pipelined_sk_kernel (the same body, sdpa_fwd_f32_pipelined.inc included with SDPA_PK_SK 1) wraps the K/V walk in a loop over the PIECES of a workgroup's run
    of tile steps.  The piece loop must not cost the steady-state loop anything: same MFMA, LDS-read, DMA and
    exp2 counts as the classic instantiation's, and no scratch traffic inside it (a back-edge around a
    full-register-file loop is exactly what made hipcc spill there in round 3).  (dk = 256 has no stream-K
    instantiation for that very reason: with its 128-register Q fragment the piece loop does spill.)"""
    classic = main_loop_mix(kernel_lines(f32_asm, "fused_pipelined_kernelILi%dELi%dELi0ELi0EE" % (dk, dv)))
    sk = main_loop_mix(kernel_lines(f32_asm, "fused_pipelined_sk_kernelILi%dELi%dEE" % (dk, dv)))
    assert sum(v for k, v in sk.items() if k.startswith("scratch_")) == 0, sk
    for op in ("v_mfma_f32_32x32x2_f32", "ds_read_b128", "ds_read_b64", "v_exp_f32", "s_barrier", "v_accvgpr_read_b32",
               "v_accvgpr_write_b32"):
        assert sk[op] == classic[op], (op, sk[op], classic[op])
    # the DMA pieces of the steady state (the cold ragged-tile block may be laid out inside the loop: <= 2x)
    assert classic["global_load_lds_dwordx4"] <= sk["global_load_lds_dwordx4"] <= 2 * classic["global_load_lds_dwordx4"], sk


def _m0_uses_outside_asm(k):
    """instructions that mention M0 outside an inline-asm block (;;#ASMSTART ... ;;#ASMEND)"""
    inside, hits = False, []
    for l in k:
        if "#ASMSTART" in l:
            inside = True
        elif "#ASMEND" in l:
            inside = False
        elif not inside and re.search(r"\bm0\b", l) and re.match(r"^\t[a-z]", l):
            hits.append(l.strip())
    return hits


def test_nothing_but_the_dma_asm_touches_m0(f32_asm, bf16_asm):
    """The LDS-DMA statements write M0 (the DMA's LDS destination) and do not restore it (only the bf16 general kernel's
    does): the asm lists M0 as clobbered, so hipcc may not keep a value of its own there.  What makes that SAFE rather
    than assumed: in every kernel with such a statement, no compiler-generated instruction reads
    or writes M0 at all -- there is nothing the DMA's leftover address could be mistaken for.  A compiler that starts
    using M0 in these kernels (LDS-direct loads, s_movrel, GWS) trips this test, and build() with it."""
    names = [(f32_asm, r"fused_pipelined_(sk_|stream_)?kernelILi\d+ELi\d+E"),
             (bf16_asm, r"fused_bf16_(tandem|tandem_stream|duo|pipe)_kernelI")]
    seen = 0
    for lines, pat in names:
        for i, l in enumerate(lines):
            m = re.match(r"^(_ZN4sdpa\S*):", l)
            if not m or not re.search(pat, m.group(1)):
                continue
            end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
            k = lines[i:end]
            if not any("global_load_lds" in x for x in k):
                continue
            seen += 1
            hits = _m0_uses_outside_asm(k)
            assert not hits, "%s: compiler-generated M0 use beside the DMA asm: %s" % (m.group(1), hits[:4])
    assert seen >= 12, seen          # 7 + 5 fp32 instantiations and the bf16 kernels were looked at
