#!/usr/bin/env python3
"""prof_timeline.py <raw marks> <grid>: per-workgroup timeline of the dominant kernel from GK_KERNEL_PROF=/abs/path marks
(8 x u64 per row group: start, cleared, phase-1 end, staged, bounds, formulas, outputs, -).  The cycle counters differ from CU
to CU (workgroups w, w + 32, w + 64 of an XCD share one): only differences within a workgroup, or between CU mates, are used."""
import sys
import numpy as np
pr = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
grid = int(sys.argv[2])
pr = pr[pr[:, 0] > 0]
ng = len(pr); per_xcd = (ng + 7) // 8; nbx = grid // 8
life = {}; hist = np.zeros(4)
for x in range(8):
    g = pr[x * per_xcd:min(ng, (x + 1) * per_xcd)]
    for w in range(min(nbx, len(g))):
        idx = list(range(w, len(g), nbx))
        life.setdefault(len(idx), []).append((w * 3 // nbx, g[idx[-1], 6] - g[idx[0], 0]))
    for c in range(nbx // 3):
        ws = [c, c + nbx // 3, c + 2 * (nbx // 3)]
        if max(ws) >= len(g): continue
        t0 = min(g[w, 0] for w in ws); t1 = max(g[list(range(w, len(g), nbx))[-1], 6] for w in ws)
        if t1 - t0 > 1000000: continue   # (not CU mates after all)
        ev = []
        for w in ws:
            for i in range(w, len(g), nbx): ev += [(g[i, 1], 1), (g[i, 2], -1)]
        ev.sort(); k = 0; last = t0
        for t, d in ev: hist[k] += t - last; last = t; k += d
        hist[0] += t1 - last
for n, v in sorted(life.items()):
    v = np.array(v)
    print("workgroups with %d groups: %d, lifetime mean %d clocks (%d .. %d); by dispatch third %s" % (n, len(v), v[:, 1].mean(), v[:, 1].min(), v[:, 1].max(),
          [int(v[v[:, 0] == k, 1].mean()) if (v[:, 0] == k).any() else None for k in range(3)]))
for r in range((per_xcd + nbx - 1) // nbx):
    d = np.concatenate([(lambda s: s[:, 6] - s[:, 0])(pr[x * per_xcd:min(ng, (x + 1) * per_xcd)][r * nbx:(r + 1) * nbx]) for x in range(8)])
    print("round %d: %d groups, mean %d clocks" % (r, len(d), d.mean()))
print("share of a CU's time with k of its 3 workgroups in phase 1:", (hist / max(hist.sum(), 1)).round(3).tolist())
