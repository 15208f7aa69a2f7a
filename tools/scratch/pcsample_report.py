#!/usr/bin/env python3
"""pcsample_report.py <samples> [lib-substring]: histogram of sampled PCs by symbol (nm) for the mapped objects."""
import bisect, collections, subprocess, sys
lines = open(sys.argv[1]).read().split('\n')
k = lines.index('MAPS')
stacks = [[int(y, 16) for y in x.split()] for x in lines[:k] if x]
import os
own = os.environ.get('PCSAMPLE_OWN', '')
pcs = []
for st in stacks:
    pcs.append(st)

maps = []
for l in lines[k + 1:]:
    p = l.split()
    if len(p) >= 6:
        a, b = [int(x, 16) for x in p[0].split('-')]
        maps.append((a, b, int(p[2], 16), p[5]))
syms = {}
def table(path):
    if path not in syms:
        out = subprocess.run(['nm', '-C', '--defined-only', '-n', path], capture_output=True, text=True).stdout
        out += subprocess.run(['nm', '-C', '-D', '--defined-only', '-n', path], capture_output=True, text=True).stdout
        t = sorted({(int(l.split(None, 2)[0], 16), l.split(None, 2)[2]) for l in out.split('\n') if len(l.split(None, 2)) == 3 and l.split(None, 2)[1] in 'tTwW'})
        syms[path] = ([a for a, _ in t], [s for _, s in t])
    return syms[path]
h = collections.Counter(); byobj = collections.Counter()
def resolve(pc):
    for a, b, off, path in maps:
        if a <= pc < b:
            addrs, names = table(path)
            i = bisect.bisect_right(addrs, pc - a + off) - 1
            return path.split('/')[-1], (names[i][:110] if i >= 0 else '?')
    return '?', hex(pc)
for st in pcs:
    o, s = resolve(st[0])
    byobj[o] += 1
    if own and own not in o:   # attribute to the first frame inside the object of interest
        for pc in st[1:]:
            if not pc: break
            o2, s2 = resolve(pc)
            if own in o2:
                s = s2 + '   <- ' + o + ':' + s[:40]; o = o2
                break
    h[(o, s)] += 1
print(len(pcs), 'samples'); print(byobj.most_common(8))
for (o, s), c in h.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 45):
    print('%6.2f%% %6d  %s  %s' % (100.0 * c / len(pcs), c, o, s))
