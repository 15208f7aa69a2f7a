#!/usr/bin/env python3
"""Does the 10 M-object sweep's kernel time depend on what the process allocated before?  (round 6: the kernel has two levels from
process to process, 0.406-0.413 and 0.434-0.458 ms; the driver-style bench, which builds the 10 M table after three others, lands on the
upper one more often than the lean run, which builds it first.)
usage: alloc_history_probe.py fresh|after [n_big]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gatekeeper_amd import driver as D, synth
mode = sys.argv[1]
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
fx = synth.load_fixtures()
drv = D.Driver(device=0)
client = D.Client(drv)
for t in synth.psp_templates(fx): client.AddTemplate(t)
for k in synth.audit_constraints(): client.AddConstraint(k)
nss = synth.gen_namespaces()
def table_of(n, start=0):
    b = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=start, namespaces=nss)
    return b, drv.engine.create_table_native(b.reviews, n, keep_docs=False, resident=True, pruned=True)
def kernel_ms(table, steps=20):
    for _ in range(5): table.launch()
    table.eval(download=False, collect_only=True)
    for _ in range(steps): table.launch(kernel_only=True)
    return table.eval(download=False, collect_only=True).fast_kernel_ms
if mode == "after":   # what the default bench line does first: a 1 M-object table (kept), a second one (freed), smaller ones (freed)
    b1, t1 = table_of(1000000)
    k1 = kernel_ms(t1)
    b2, t2 = table_of(1000000); t2.free(); del b2
    b3, t3 = table_of(100000); kernel_ms(t3); t3.free(); del b3
    b4, t4 = table_of(200000); kernel_ms(t4); t4.free(); del b4
    print("1 M-object table first: kernel %.4f ms" % k1)
bb, tb = table_of(n_big)
print("%s: %d objects, kernel %.4f ms" % (mode, n_big, kernel_ms(tb)))
