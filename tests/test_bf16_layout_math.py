"""CPU: index math of the bf16 kernels' LDS staging (csrc/sdpa_fwd_bf16.hip, sdpa_fwd_bf16_tandem.inc), restated in Python.

Two image families (include/sdpa_hip.h): ROW images (dv <= 256: the duo / pipe kernels swizzle on the SOURCE side of
their LDS-DMA pieces) and TILED images (dv > 256, round 6: the converters write every tile in the LDS buffer's byte order,
the tandem kernel's pieces are lane-linear at both ends).  Both must fill the LDS buffers identically: the fragment
reads are the same.  Checks that the DMA placement (one 1-KiB piece per wave-instruction), the
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
def test_k_tile_placement_and_reads_row_images(DK):
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


def test_vt_tile_placement_key_order_and_reads_row_images():
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


# ---- tiled images (dv > 256): what the converters write is what the tandem kernel's LDS buffers must hold ----------------------
def tiled_k_position(r, c, DK):
    """byte offset inside the K image of 16-byte chunk c of row r: chunk position c ^ (r & swz) of the row"""
    KCH = DK // 8
    SWZ = 15 if KCH >= 16 else KCH - 1
    return r * DK * 2 + ((c ^ (r & SWZ)) << 4)


def tiled_vt_position(tile, col, key, dvp):
    """element offset inside the Vt image of key `key` (0..31 of tile `tile`) of column `col`"""
    p = kvpos(key)
    block = tile * (dvp // 512) + col // 512
    row = col % 512
    return block * 512 * 32 + row * 32 + ((((p >> 3) ^ ((row >> 2) & 3)) << 3) | (p & 7))


@pytest.mark.parametrize("DK", [64, 128, 256, 512])
def test_tiled_k_image_fills_the_lds_buffer_the_fragment_reads_expect(DK):
    KCH = DK // 8
    KPW = (32 * KCH // 64) // 4
    SWZ = 15 if KCH >= 16 else KCH - 1
    KBIAS = 4096 if KPW == 8 else 0
    # what sits at every 16-byte slot of a tile of the image: (row, original chunk)
    image = {}
    for r in range(32):
        for c in range(KCH):
            image[tiled_k_position(r, c, DK) // 16] = (r, c)
    assert len(image) == 32 * KCH
    # the kernel's pieces: base = tile + wave * KPW * 1024 + KBIAS, immediate = j * 1024 - KBIAS, lane * 16; M0 likewise
    lds = {}
    for wave in range(4):
        for j in range(KPW):
            imm = j * 1024 - KBIAS
            assert -4096 <= imm <= 4095, "the instruction's signed 13-bit offset"
            for lane in range(64):
                src = wave * KPW * 1024 + KBIAS + imm + lane * 16
                dst = wave * KPW * 1024 + KBIAS + imm + lane * 16
                assert dst // 16 not in lds
                lds[dst // 16] = image[src // 16]
    assert len(lds) == 32 * KCH
    NKS = DK // 16
    NKA = min(NKS, 8)
    for li in range(32):
        for hi in range(2):
            for ks in range(NKS):
                addr = li * DK * 2 + (((2 * (ks % NKA) + hi) ^ (li & SWZ)) << 4) + (ks // NKA) * 256
                assert lds[addr // 16] == (li, 2 * ks + hi)


@pytest.mark.parametrize("dvp,chunk", [(512, 0), (1024, 1)])
def test_tiled_vt_image_fills_the_lds_buffer_the_fragment_reads_expect(dvp, chunk):
    tile = 3
    image = {}
    for col in range(dvp):
        for key in range(32):
            image[tiled_vt_position(tile, col, key, dvp)] = (col, key)
    assert len(image) == dvp * 32 and min(image) == tile * dvp * 32 and max(image) == (tile + 1) * dvp * 32 - 1
    # the kernel: vimg = Vt + tile * (n_chunks * VTILE) + chunk * VTILE + wave * 8 KiB + 4096; pieces at -4096 .. 3072
    n_chunks = dvp // 512
    lds = {}
    for wave in range(4):
        for j in range(8):
            imm = j * 1024 - 4096
            for lane in range(64):
                src_byte = (tile * n_chunks + chunk) * 512 * 32 * 2 + wave * 8192 + 4096 + imm + lane * 16
                dst_byte = wave * 8192 + 4096 + imm + lane * 16
                lds[dst_byte // 16] = [image[src_byte // 2 + e] for e in range(8)]
    assert len(lds) == 512 * 64 // 16
    for role in range(2):
        for li in range(32):
            for hi in range(2):
                for h in range(2):
                    vaddr = role * 8 * 2048 + li * 64 + (((2 * h + hi) ^ ((li >> 2) & 3)) << 4)
                    for tt in range(8):
                        got = lds[(vaddr + tt * 2048) // 16]
                        # the eight keys of this lane's P.V operand: accumulator registers 8h .. 8h+7 of the score tile
                        assert got == [(512 * chunk + 256 * role + 32 * tt + li, crow16(8 * h + j, hi)) for j in range(8)]
