#!/usr/bin/env python3
"""Bank-conflict model of the AWGN wave kernel's two blocked tap-gradient phases (csrc/vaeq_awgn_wave.hip, P4a dL/dh and P5 dL/dw).

For a lane -> (group, part) mapping and a part stride it counts, per loop trip, the LDS passes of every 8-byte operand read: a read of
64 lanes is served as two half-waves of 32; within a half-wave the lanes that name the SAME dword are one access (broadcast), different
dwords on the same of the 64 banks serialise.  passes(read) = sum over the two half-waves of the maximum bank multiplicity.
Usage: python tools/lds_bank_model.py [B] [M]
"""
import sys
from collections import defaultdict


def wave_lph(n):
    lph = (n + 3) // 4 + 1
    while (lph & 31) not in (8, 24):
        lph += 1
    return lph


def passes(addrs):
    """addrs: list over the 64 lanes of a float2 index (None = idle lane: shadows a working lane of its half-wave)."""
    total = 0
    for h in range(2):
        banks = defaultdict(set)
        half = addrs[32 * h:32 * h + 32]
        base = next((a for a in half if a is not None), addrs[0])    # an idle lane shadows a working lane of its own half-wave
        for a in half:
            a = base if a is None else a
            for d in (2 * a, 2 * a + 1):
                banks[d % 64].add(d)
        total += max(len(s) for s in banks.values())
    return total


def model(B, M, lane_map_h, lane_map_w, nit_h=None, nit_w=None, gy_planar=False, verbose=True):
    mh = M // 2
    Mh = 2 * mh
    L = 2 * B
    nm = L - Mh
    Lph = wave_lph(2 * B + M - 1)
    Uph = B // 2 + 1
    NG0, NG1 = (mh + 1 + 3) // 4, (mh + 3) // 4
    NGH, NGW = NG0 + NG1, (M + 3) // 4
    # LDS float2 offsets of the arrays (bytes / 8), as awgn_wave_layout lays them out
    X = 0
    E = X + 4 * Lph
    U = E + 4 * Lph
    # ---- P4a
    T = nm >> 1
    Tm = (T + 1) >> 1
    PH = max(p for _, p in lane_map_h if p is not None) + 1
    nit = nit_h or (Tm + PH - 1) // PH
    assert nit * PH >= Tm, (nit, PH, Tm)
    reads_h = []
    for rd in range(7):
        addrs = []
        for g, p in lane_map_h:
            if g is None:
                addrs.append(None)
                continue
            par = 1 if g >= NG0 else 0
            a0 = 4 * (g - NG0 if par else g)
            m0 = p * nit
            ceA, ceB, n0 = par + Mh, par + Mh + 2, mh - a0 - 3
            d = n0 & 1
            eA = E + (ceA & 3) * Lph + (ceA >> 2) + m0
            eB = E + (ceB & 3) * Lph + (ceB >> 2) + m0
            uE = U + d * Uph + (n0 >> 1) + m0
            uO = U + (d ^ 1) * Uph + (n0 >> 1) + d + m0
            addrs.append([eA, eB, uE, uO, uE + 1, uO + 1, uE + 2][rd])
        reads_h.append(passes(addrs))
    # ---- P5
    Bp = B >> 1
    PW = max(p for _, p in lane_map_w if p is not None) + 1
    nitw = nit_w or (Bp + PW - 1) // PW
    assert nitw * PW >= Bp
    reads_w = []
    for rd in range(8):
        addrs = []
        for g, p in lane_map_w:
            if g is None:
                addrs.append(None)
                continue
            m0 = p * nitw
            xw = X + g + m0
            if gy_planar:
                gy = [U + m0, U + Bp + 8 + m0]      # two planar arrays (even / odd symbols)
            else:
                gy = [U + 2 * m0, U + 2 * m0 + 1]
            addrs.append([gy[0], gy[1], xw, xw + Lph, xw + 2 * Lph, xw + 3 * Lph, xw + 1, xw + Lph + 1][rd])
        reads_w.append(passes(addrs))
    if verbose:
        print(f"  dL/dh: {PH} parts x {nit} trips, passes per read {reads_h} -> {sum(reads_h)} per trip, {sum(reads_h) * nit} per step (ideal {14 * nit})")
        print(f"  dL/dw: {PW} parts x {nitw} trips, passes per read {reads_w} -> {sum(reads_w)} per trip, {sum(reads_w) * nitw} per step (ideal {16 * nitw})")
    return sum(reads_h) * nit, sum(reads_w) * nitw


def group_major(NG, P):
    out = []
    for gl in range(64):
        g, p = gl // P, gl % P
        out.append((g, p) if g < NG else (None, None))
    return out


def half_split(NG, P):
    """parts 0..P/2-1 in the first half-wave, the rest in the second; group-minor inside a part"""
    out = [(None, None)] * 64
    hp = P // 2
    for p in range(P):
        h, pp = divmod(p, hp)
        for g in range(NG):
            out[32 * h + pp * NG + g] = (g, p)
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 350
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    mh = M // 2
    NGH, NGW = (mh + 4) // 4 + (mh + 3) // 4, (M + 3) // 4
    print(f"B = {B}, M = {M}: Lph = {wave_lph(2 * B + M - 1)}, groups dL/dh {NGH}, dL/dw {NGW}")
    print("shipped mapping (lane = group * parts + part, parts = 64 / groups):")
    model(B, M, group_major(NGH, 64 // NGH), group_major(NGW, 64 // NGW))
    print("same, gy planar:")
    model(B, M, group_major(NGH, 64 // NGH), group_major(NGW, 64 // NGW), gy_planar=True)
    for nit_h, nit_w in ((None, None), (24, 24), (22, 24), (22, 22), (23, 24)):
        for planar in (False, True):
            print(f"8 parts, 4 per half-wave, trips {nit_h}/{nit_w}, gy planar {planar}:")
            try:
                model(B, M, half_split(NGH, 8), half_split(NGW, 8), nit_h, nit_w, planar)
            except AssertionError as e:
                print("  n/a", e)
