# Simulate the wide kernel's DMA placement and fragment reads (index math only).
import numpy as np
def kvpos(j): return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1)
def crow16(r, hi): return (r & 3) + 8 * (r >> 2) + 4 * hi

def check_k(DK):
    KCH = DK // 8; KTILE_B = 32 * DK * 2
    KPW = (32 * KCH // 64) // 4; RPP = max(64 // KCH, 1); SWZ = 15 if KCH >= 16 else KCH - 1
    # global K tile: element id = row*DK + col ; bytes: 2 per element. Represent 16-B chunks by (row, chunk)
    lds = {}  # lds 16B slot index -> (row, chunk)
    for wave in range(4):
        for j in range(KPW):
            row0 = (wave * KPW + j) * RPP
            swz = (row0 & SWZ) << 4
            for lane in range(64):
                klane = (lane // KCH) * DK * 2 + (((lane % KCH) ^ ((lane // KCH) & SWZ)) << 4)
                off = row0 * DK * 2 + (klane ^ swz)       # byte offset in the tile
                row, chunk = off // (DK * 2), (off % (DK * 2)) // 16
                slot = ((wave * KPW + j) * 1024 + lane * 16) // 16
                assert slot not in lds
                lds[slot] = (row, chunk)
    assert len(lds) == KTILE_B // 16
    NKS = DK // 16; NKA = min(NKS, 8)
    for li in range(32):
        for hi in range(2):
            for ks in range(NKS):
                u = ks % NKA
                kaddr = li * DK * 2 + (((2 * u + hi) ^ (li & SWZ)) << 4)
                addr = kaddr + (ks // NKA) * 256
                got = lds[addr // 16]
                want = (li, 2 * ks + hi)
                assert got == want, (DK, li, hi, ks, got, want)
    bad = False
    for ks in range(NKS):
        for g in range(4):
            banks = set()
            for lane in range(16 * g, 16 * g + 16):
                li, hi = lane & 31, lane >> 5
                u = ks % NKA
                addr = li * DK * 2 + (((2 * u + hi) ^ (li & SWZ)) << 4) + (ks // NKA) * 256
                b = (addr // 4) % 64
                for q in range(4): banks.add((b + q) % 64)
            if len(banks) != 64: bad = True
    print("K ok", DK, "bank-conflicts" if bad else "conflict-free")

def check_v():
    DVC = 512; ldvt = 96  # bf16 elements per Vt row in the image (multiple of 32)
    tile = 1  # kv base = 32
    # image: Vt[row][pos] holds V[kv = kvpos^-1(pos) = kvpos(pos)][col=row]
    lds = {}
    VPW = 8
    for wave in range(4):
        for j in range(VPW):
            for lane in range(64):
                vlane = (lane >> 2) * ldvt * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4)
                base = ((wave * VPW + j) * 16) * ldvt * 2 + tile * 32 * 2
                off = base + vlane                       # byte offset in the image
                row, posb = off // (ldvt * 2), off % (ldvt * 2)
                slot = ((wave * VPW + j) * 1024 + lane * 16) // 16
                assert slot not in lds
                lds[slot] = (row, posb // 2)             # row, first element position (8 elements)
    assert len(lds) == 512 * 64 // 16
    for li in range(32):
        for hi in range(2):
            for h in range(2):
                vaddr = li * 64 + (((2 * h + hi) ^ ((li >> 2) & 3)) << 4)
                for tt in range(16):
                    row, pos0 = lds[(vaddr + tt * 2048) // 16]
                    assert row == 32 * tt + li
                    kvs = [kvpos(p) for p in range(pos0, pos0 + 8)]
                    # positions pos0..pos0+7 of the image row hold keys kvpos(p) (involution)
                    want = [tile * 32 + crow16(8 * h + jj, hi) for jj in range(8)]
                    assert kvs == want, (li, hi, h, tt, kvs, want)
    for h in range(2):
        for tt in range(16):
            for g in range(4):
                banks = set()
                for lane in range(16 * g, 16 * g + 16):
                    li, hi = lane & 31, lane >> 5
                    addr = li * 64 + (((2 * h + hi) ^ ((li >> 2) & 3)) << 4) + tt * 2048
                    b = (addr // 4) % 64
                    for q in range(4): banks.add((b + q) % 64)
                assert len(banks) == 64, (h, tt, g, len(banks))
    print("V ok")

for DK in (64, 128, 256, 512): check_k(DK)
check_v()
