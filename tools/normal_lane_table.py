#!/usr/bin/env python3
"""Lane -> chunk table of ba_normal_kernel (csrc/ba_normal.hpp).

The upper triangle (p <= q < NA) is cut row by row into chunks (p; q0 .. q0+CH-1), one per lane.  All lanes
read the same LDS row, lane t at slots p_t and q0_t + j; ds_read_b128 serves a wave in four 16-lane groups
({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32) over 64 banks = 16 slots, so two lanes of a group
collide when their slots differ by exactly 16.  This script searches an assignment of chunks to lanes without
such pairs (for p and for q0; the shift j is common to all lanes) and prints it as C tables."""
import random
import sys

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[x + 32 for x in g] for g in GROUPS]


def chunks(na, ch):
    return [(p, q0) for p in range(na) for q0 in range(p, na, ch)]


def conflicts(assign):
    n = 0
    for g in GROUPS:
        for key in (0, 1):
            vals = {(assign[l] or (0, 0))[key] for l in g}   # an idle lane still reads: slot 0 (+ j)
            n += sum(1 for v in vals if v + 16 in vals)
    return n


def search(na, ch, seed=0):
    rng = random.Random(seed)
    cs = chunks(na, ch)
    assert len(cs) <= 64
    assign = cs + [None] * (64 - len(cs))
    best = conflicts(assign)
    for _ in range(200000):
        if best == 0:
            break
        a, b = rng.randrange(64), rng.randrange(64)
        assign[a], assign[b] = assign[b], assign[a]
        c = conflicts(assign)
        if c <= best:
            best = c
        else:
            assign[a], assign[b] = assign[b], assign[a]
    return assign, best


if __name__ == "__main__":
    for na, ch in ((22, 5), (16, 3)):
        assign, left = search(na, ch)
        print(f"// NA = {na}, CH = {ch}: {len(chunks(na, ch))} chunks, {left} slot pairs 16 apart left (sequential order: {conflicts(chunks(na, ch) + [None] * (64 - len(chunks(na, ch))))})")
        print(f"constexpr unsigned char NORMAL_P_{na}[64] = {{" + ", ".join(str(a[0]) if a else "255" for a in assign) + "};")
        print(f"constexpr unsigned char NORMAL_Q_{na}[64] = {{" + ", ".join(str(a[1]) if a else "255" for a in assign) + "};")
