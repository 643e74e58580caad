"""CPU: index math of the bf16 wide kernel's LDS staging (csrc/sdpa_fwd_bf16.hip), restated in Python.

Checks that the DMA placement (source-side XOR swizzles, one 1-KiB piece per wave-instruction), the
fragment read addresses and the kvpos() key order of the Vt image are mutually consistent -- every
fragment read lands on the (row, chunk) it is meant to, the eight keys of a P.V operand are the ones
the score tile's accumulator registers hold -- and that the reads are bank-conflict free per 16-lane
group.  It does not run the kernel; tests/test_gpu_bf16.py does."""
import pytest


def kvpos(j):
    return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1)


def crow16(r, hi):
    return (r & 3) + 8 * (r >> 2) + 4 * hi


@pytest.mark.parametrize("DK", [64, 128, 256, 512])
def test_k_tile_placement_and_reads(DK):
    KCH = DK // 8
    KPW = (32 * KCH // 64) // 4
    RPP = max(64 // KCH, 1)
    SWZ = 15 if KCH >= 16 else KCH - 1
    lds = {}
    for wave in range(4):
        for j in range(KPW):
            row0 = (wave * KPW + j) * RPP
            swz = (row0 & SWZ) << 4
            for lane in range(64):
                klane = (lane // KCH) * DK * 2 + (((lane % KCH) ^ ((lane // KCH) & SWZ)) << 4)
                off = row0 * DK * 2 + (klane ^ swz)
                slot = ((wave * KPW + j) * 1024 + lane * 16) // 16
                assert slot not in lds
                lds[slot] = (off // (DK * 2), (off % (DK * 2)) // 16)
    assert len(lds) == 32 * DK * 2 // 16
    NKS = DK // 16
    NKA = min(NKS, 8)
    for li in range(32):
        for hi in range(2):
            for ks in range(NKS):
                addr = li * DK * 2 + (((2 * (ks % NKA) + hi) ^ (li & SWZ)) << 4) + (ks // NKA) * 256
                assert lds[addr // 16] == (li, 2 * ks + hi)
    if DK >= 128:      # 64-float rows and wider: every 16-lane group of a ds_read_b128 covers all 64 banks
        for ks in range(NKS):
            for g in range(4):
                banks = set()
                for lane in range(16 * g, 16 * g + 16):
                    li, hi = lane & 31, lane >> 5
                    addr = li * DK * 2 + (((2 * (ks % NKA) + hi) ^ (li & SWZ)) << 4) + (ks // NKA) * 256
                    banks.update(((addr // 4) + q) % 64 for q in range(4))
                assert len(banks) == 64


def test_vt_tile_placement_key_order_and_reads():
    ldvt, tile = 96, 1
    lds = {}
    for wave in range(4):
        for j in range(8):
            for lane in range(64):
                vlane = (lane >> 2) * ldvt * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4)
                off = ((wave * 8 + j) * 16) * ldvt * 2 + tile * 32 * 2 + vlane
                slot = ((wave * 8 + j) * 1024 + lane * 16) // 16
                assert slot not in lds
                lds[slot] = (off // (ldvt * 2), (off % (ldvt * 2)) // 2)
    assert len(lds) == 512 * 64 // 16
    for li in range(32):
        for hi in range(2):
            for h in range(2):
                vaddr = li * 64 + (((2 * h + hi) ^ ((li >> 2) & 3)) << 4)
                for tt in range(16):
                    row, pos0 = lds[(vaddr + tt * 2048) // 16]
                    assert row == 32 * tt + li
                    # positions pos0..pos0+7 of the image row hold keys kvpos(p); the P operand's
                    # k-slot j is accumulator register 8h+j of the score tile = key row crow16(8h+j, hi)
                    assert [kvpos(p) for p in range(pos0, pos0 + 8)] == [tile * 32 + crow16(8 * h + j, hi) for j in range(8)]
    for h in range(2):
        for tt in range(16):
            for g in range(4):
                banks = set()
                for lane in range(16 * g, 16 * g + 16):
                    li, hi = lane & 31, lane >> 5
                    addr = li * 64 + (((2 * h + hi) ^ ((li >> 2) & 3)) << 4) + tt * 2048
                    banks.update(((addr // 4) + q) % 64 for q in range(4))
                assert len(banks) == 64
