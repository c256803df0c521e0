"""Bank-conflict simulation of the ds_read_b128 fragment reads of the 256-wide forward kernels' WEIGHT stage images
(64-B rows, four 16-B chunks, chunk index XOR-swizzled by row bits) for candidate swizzles and row orders.
ds_read_b128 is served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS): a group is conflict-free when its 16
lanes touch 16 distinct 16-B bank slots.  `orig` = 16 consecutive rows per tile (the LDS-staged epilogues), `perm` = the
rows 0-3, 8-11, 16-19, 24-27 (+4 for odd tiles) the register epilogue reads (csrc/conv.hip, epilogue_direct)."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]
ORDERS = {"orig": lambda j, m: 16 * j + m, "perm": lambda j, m: 32 * (j >> 1) + 8 * (m >> 2) + 4 * (j & 1) + (m & 3)}
SWIZZLES = {"bits 1..2 (SLN_SWZH)": lambda r: (r >> 1) & 3, "bits 2..3": lambda r: (r >> 2) & 3,
            "bits 1 and 3 (SLN_SWZW)": lambda r: ((r >> 1) & 1) | (((r >> 3) & 1) << 1)}


def worst(order, swz):
    w = 0
    for j in range(4):
        for g in GROUPS:
            slots = {}
            for lane in g:
                row = order(j, lane & 15)
                slot = ((row * 64 + (((lane >> 4) ^ swz(row)) * 16)) // 16) % 16
                slots[slot] = slots.get(slot, 0) + 1
            w = max(w, max(slots.values()))
    return w


if __name__ == "__main__":
    for name, swz in SWIZZLES.items():
        print("%-26s %s" % (name, "  ".join("%s: %d-way" % (k, worst(o, swz)) for k, o in ORDERS.items())))
