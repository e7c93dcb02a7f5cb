"""0-1-principle check of the 99-comparator median-of-25 network used by k_median5_c2 (flow_kernels.hip): every one of\nthe 2^25 binary inputs must come out as its majority bit. Run: python tools/verify_median_network.py (~20 s, ~1 GB)."""
import numpy as np
NET = """
0 1 3 4 2 4 2 3 6 7 5 7 5 6 9 10 8 10 8 9 12 13 11 13 11 12 15 16 14 16 14 15 18 19 17 19 17 18 21 22 20 22 20 21 23 24
2 5 3 6 0 6 0 3 4 7 1 7 1 4 11 14 8 14 8 11 12 15 9 15 9 12 13 16 10 16 10 13 20 23 17 23 17 20 21 24 18 24 18 21 19 22
8 17 9 18 0 18 0 9 10 19 1 19 1 10 11 20 2 20 2 11 12 21 3 21 3 12 13 22 4 22 4 13 14 23 5 23 5 14 15 24 6 24 6 15 7 16
7 19 13 21 15 23 7 13 7 15 1 9 3 11 5 17 11 17 9 17 4 10 6 12 7 14 4 6 4 7 12 14 10 14 6 7 10 12 6 10 6 17 12 17 7 17 7 10
12 18 7 12 10 18 12 20 10 20 10 12
"""
pairs = np.array(NET.split(), dtype=int).reshape(-1, 2)
print(len(pairs), "comparators")
N = 1 << 25
n = np.arange(N, dtype=np.uint32)
# popcount
v = n - ((n >> 1) & 0x55555555); v = (v & 0x33333333) + ((v >> 2) & 0x33333333); pc = (((v + (v >> 4)) & 0x0F0F0F0F) * 0x01010101) >> 24
maj = np.packbits((pc >= 13).astype(np.uint8), bitorder='little').view(np.uint64)
wires = [np.packbits(((n >> i) & 1).astype(np.uint8), bitorder='little').view(np.uint64) for i in range(25)]
# The kernel runs the 22 runs of three comparators that sort three wires as sort3 (min3 / med3 / max3); group them
# the same way here (a run (b,c),(a,c),(a,b) over wires a<b<c) and apply the 3-sort as one operation.
ops, i = [], 0
while i < len(pairs):
    if i + 2 < len(pairs):
        (p0, p1), (q0, q1), (r0, r1) = pairs[i], pairs[i + 1], pairs[i + 2]
        ws = sorted({p0, p1, q0, q1, r0, r1})
        if len(ws) == 3 and (p0, p1) == (ws[1], ws[2]) and (q0, q1) == (ws[0], ws[2]) and (r0, r1) == (ws[0], ws[1]):
            ops.append(tuple(ws)); i += 3; continue
    ops.append(tuple(pairs[i])); i += 1
print(sum(len(o) == 3 for o in ops), "sort3 +", sum(len(o) == 2 for o in ops), "compare-exchange")
for o in ops:
    if len(o) == 2:
        a, b = o
        lo = wires[a] & wires[b]; hi = wires[a] | wires[b]
        wires[a], wires[b] = lo, hi
    else:
        a, b, c = o
        lo = wires[a] & wires[b] & wires[c]; hi = wires[a] | wires[b] | wires[c]
        md = (wires[a] & wires[b]) | (wires[a] & wires[c]) | (wires[b] & wires[c])
        wires[a], wires[b], wires[c] = lo, md, hi
ok = np.array_equal(wires[12], maj)
print("median network correct for all 2^25 0/1 inputs:", ok)
if not ok:
    bad = np.count_nonzero(wires[12] != maj); print("bad words", bad)
