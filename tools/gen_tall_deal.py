"""Generates the block dealings of gram_tall_kernel<NBC> (csrc/gram_tall.hip: the `tw_*` constexpr tables and TALL_BLOCKS): the upper triangle
of an NBC x NBC grid of 16 x 16 blocks dealt to the four waves of a workgroup so that the waves' MFMA counts (3 per diagonal block, 4 per
other) are level and each wave needs few distinct operand rows / columns from LDS.  NBC = 8 is the hand-made dealing of round 5; 5, 6, 7 come
from a small annealing search (python tools/gen_tall_deal.py prints the C++ that was pasted into the source)."""
DEALS = {
    5: [[(2, 4), (3, 4), (4, 4)], [(0, 0), (0, 2), (1, 2), (2, 2)], [(0, 3), (1, 3), (2, 3), (3, 3)], [(0, 1), (1, 1), (0, 4), (1, 4)]],
    6: [[(0, 5), (1, 5), (2, 5), (3, 5), (4, 5)], [(0, 4), (1, 4), (2, 4), (3, 4), (4, 4)], [(0, 1), (0, 3), (1, 3), (2, 3), (3, 3)],
        [(0, 0), (1, 1), (0, 2), (1, 2), (2, 2), (5, 5)]],
    7: [[(0, 0), (0, 2), (1, 2), (2, 2), (0, 6), (1, 6), (2, 6)], [(3, 5), (4, 5), (5, 5), (3, 6), (4, 6), (5, 6), (6, 6)],
        [(0, 3), (1, 3), (2, 3), (3, 3), (0, 5), (1, 5), (2, 5)], [(0, 1), (1, 1), (0, 4), (1, 4), (2, 4), (3, 4), (4, 4)]],
    8: [[(0, 5), (1, 5), (2, 5), (0, 6), (1, 6), (2, 6), (0, 7), (1, 7), (2, 7)], [(0, 3), (1, 3), (2, 3), (3, 3), (0, 4), (1, 4), (2, 4), (3, 4), (4, 4)],
        [(0, 0), (0, 1), (1, 1), (0, 2), (1, 2), (2, 2), (3, 5), (3, 6), (3, 7)], [(4, 5), (5, 5), (4, 6), (5, 6), (6, 6), (4, 7), (5, 7), (6, 7), (7, 7)]],
}
MAXB, MAXR = 9, 6


def tables():
    out = {}
    for nbc, waves in DEALS.items():
        seen = sorted(b for w in waves for b in w)
        assert seen == sorted((a, b) for b in range(nbc) for a in range(b + 1)), nbc
        t = {"nblk": [], "nr": [], "nc": [], "row": [], "col": [], "blk": [], "blocks": []}
        for w in waves:
            rows, cols = sorted({b[0] for b in w}), sorted({b[1] for b in w})
            t["nblk"].append(len(w)); t["nr"].append(len(rows)); t["nc"].append(len(cols))
            t["row"].append(rows + [0] * (MAXR - len(rows))); t["col"].append(cols + [0] * (MAXR - len(cols)))
            t["blk"].append([[rows.index(a), cols.index(b)] for a, b in w] + [[0, MAXR]] * (MAXB - len(w)))      # (a padding block matches no column)
            t["blocks"].append([list(b) for b in w] + [[-1, -1]] * (MAXB - len(w)))
        out[nbc] = t
    return out


def braces(x):
    return "{" + ", ".join(braces(v) if isinstance(v, list) else str(v) for v in x) + "}"


if __name__ == "__main__":
    T = tables()
    order = [5, 6, 7, 8]
    for name, dims in (("nblk", "[4][4]"), ("nr", "[4][4]"), ("nc", "[4][4]"), ("row", "[4][4][%d]" % MAXR), ("col", "[4][4][%d]" % MAXR),
                       ("blk", "[4][4][%d][2]" % MAXB), ("blocks", "[4][4][%d][2]" % MAXB)):
        print("%s%s = %s;" % (name, dims, braces([T[n][name] for n in order])))
    for n in order:
        mf = [sum(3 if a == b else 4 for a, b in w) for w in DEALS[n]]
        print("// NBC = %d: MFMAs per pair of rotations and wave %s (needed %d, executed / needed flops %.2f)" % (n, mf, sum(mf), 4 * max(mf) * 256 * 1.0 / (16 * n * (16 * n + 1) / 2 * 4) ))
