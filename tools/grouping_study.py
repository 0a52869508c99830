#!/usr/bin/env python
"""CPU study: how much MFMA padding do other ways of forming the 16-row groups leave?  Input: gpurun_out/masks.npz
(tools/dump_masks.py).  padding = 16 * sum_groups popcount(OR of the group's masks) / sum_rows popcount(mask)."""
import numpy as np, sys
d = np.load("gpurun_out/masks.npz")
POP = np.array([bin(i).count("1") for i in range(1 << 16)], dtype=np.uint8)
def popc(x):
    x = x.astype(np.uint32)
    return POP[x & 0xFFFF].astype(np.int32) + POP[x >> 16].astype(np.int32)
def pad_of(groups_or, total_pairs):
    return 16.0 * popc(groups_or).sum() / total_pairs
def group_consecutive(m):                       # m sorted; pad to multiple of 16 with zeros
    n = len(m); k = (n + 15) // 16 * 16
    mm = np.zeros(k, np.uint32); mm[:n] = m
    return np.bitwise_or.reduce(mm.reshape(-1, 16), axis=1)
def greedy(m):
    m = m.copy(); n = len(m)
    left = np.ones(n, bool); ors = []
    pc = popc(m)
    order = np.argsort(-pc, kind="stable")
    for seed in order:
        if not left[seed]: continue
        left[seed] = False; cur = m[seed]
        for _ in range(15):
            idx = np.nonzero(left)[0]
            if len(idx) == 0: break
            grow = popc(m[idx] | cur) - int(popc(np.array([cur]))[0])
            best = idx[np.lexsort((-popc(m[idx] & cur), grow))[0]]
            left[best] = False; cur |= m[best]
        ors.append(cur)
    return np.array(ors, np.uint32)
for lvl in (1, 2, 3, 4, 5):
    sm = d[f"slotmask_{lvl}"]
    win_groups = 64 if lvl <= 3 else 16
    nwin = len(sm) // win_groups
    tot = popc(sm.reshape(-1)).sum()
    cur = pad_of(np.bitwise_or.reduce(sm, axis=1), tot)
    res = {"current": cur}
    # per-bit frequency remap like the kernel's? compare: plain sort, popcount-major sort, greedy; windows x1, x4
    for wmul in (1, 4):
        ors_plain, ors_pc, ors_gr = [], [], []
        step = win_groups * wmul
        for w0 in range(0, len(sm), step):
            m = sm[w0:w0 + step].reshape(-1); m = m[m != 0]
            if len(m) == 0: continue
            ors_plain.append(group_consecutive(np.sort(m)))
            key = (popc(m).astype(np.uint64) << 32) | m
            ors_pc.append(group_consecutive(m[np.argsort(key)]))
            if wmul == 1 and (lvl >= 2 or w0 < 40 * step): ors_gr.append((greedy(m), popc(m).sum()))
        res[f"plain_x{wmul}"] = pad_of(np.concatenate(ors_plain), tot)
        res[f"popc_major_x{wmul}"] = pad_of(np.concatenate(ors_pc), tot)
        if ors_gr:
            res["greedy_x1(sample)"] = 16.0 * sum(popc(o).sum() for o, _ in ors_gr) / sum(t for _, t in ors_gr)
    print(f"L{lvl}: " + "  ".join(f"{k} {v:.3f}" for k, v in res.items()), flush=True)

# ---- how many of the (remapped, rare-offsets-first) mask bits does the window sort have to honour?
ORDER = [13, 12, 14, 10, 16, 9, 11, 15, 17, 4, 22, 1, 7, 3, 5, 19, 25, 21, 23, 0, 2, 6, 8, 18, 20, 24, 26]
def remap(m):
    r = np.zeros_like(m)
    for i, o in enumerate(ORDER):
        r |= ((m >> np.uint32(o)) & np.uint32(1)) << np.uint32(i)
    return r
print("padding when the window sort is a stable sort on the top T bits of the remapped mask (T = 27: what ships)")
for lvl in (1, 2, 3, 4):
    sm = d[f"slotmask_{lvl}"]
    win_groups = 64 if lvl <= 3 else 16
    tot = popc(sm.reshape(-1)).sum()
    res = {}
    for T in (27, 18, 14, 12, 10, 8, 6, 0):
        ors = []
        for w0 in range(0, len(sm), win_groups):
            m = sm[w0:w0 + win_groups].reshape(-1); m = m[m != 0]
            if len(m) == 0: continue
            key = remap(m) >> np.uint32(27 - T) if T else np.zeros_like(m)
            ors.append(group_consecutive(m[np.argsort(key, kind="stable")]))
        res[T] = pad_of(np.concatenate(ors), tot)
    print(f"L{lvl}: " + "  ".join(f"T={k}: {v:.3f}" for k, v in res.items()), flush=True)
