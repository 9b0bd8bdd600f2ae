#!/usr/bin/env python3
"""round 6: the dense intra band walk (kernel_intra.h) with iterations of ONE macroblock type, simulated before building it.
A wavefront owns four rows; a row's next macroblock is ready when the row above is two columns ahead.  Cost of an iteration:
c0 (book-keeping) + c4 if any Intra4x4 macroblock runs + c16 if any Intra16x16 one does.  Policies: all ready rows (shipped);
the type of the topmost ready row; the type most ready rows have; one type only when it has two rows or more."""
import random


def sim(policy, W=80, H=45, c4=420, c16=120, c0=125, seed=1, band=4):
    rng = random.Random(seed)
    typ = [[rng.random() < 0.5 for _ in range(W)] for _ in range(H)]   # True = Intra16x16
    total = iters = 0
    for b0 in range(0, H, band):
        rows = list(range(b0, min(b0 + band, H)))
        col = [0] * len(rows)
        flip = 0
        while any(c < W for c in col):
            go = [i for i, r in enumerate(rows) if col[i] < W and (i == 0 or min(col[i] + 2, W) <= col[i - 1])]
            t16 = [i for i in go if typ[rows[i]][col[i]]]
            t4 = [i for i in go if not typ[rows[i]][col[i]]]
            if policy == "all" or not t16 or not t4:
                sel = go
            elif policy == "top":
                sel = t16 if typ[rows[go[0]]][col[go[0]]] else t4
            elif policy == "major":
                if len(t16) != len(t4):
                    sel = t16 if len(t16) > len(t4) else t4
                else:
                    sel = t16 if flip else t4
                    flip ^= 1
            else:
                sel = go if len(t16) < 2 and len(t4) < 2 else (t16 if len(t16) >= len(t4) else t4)
            has16 = any(typ[rows[i]][col[i]] for i in sel)
            has4 = any(not typ[rows[i]][col[i]] for i in sel)
            total += c0 + (c16 if has16 else 0) + (c4 if has4 else 0)
            iters += 1
            for i in sel:
                col[i] += 1
    return round(total / (W * H), 1), iters


if __name__ == "__main__":
    for p in ("all", "top", "major", "pairs"):
        print(p, sim(p))
