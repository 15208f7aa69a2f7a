#!/usr/bin/env python3
"""Where the 20-step timed region of bench.py goes beyond 20 x (sweep + totals): enqueue, the collecting call, the synchronisation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gatekeeper_amd import driver as D, synth
fx = synth.load_fixtures()
drv = D.Driver(device=0)
client = D.Client(drv)
for t in synth.psp_templates(fx): client.AddTemplate(t)
for k in synth.audit_constraints(): client.AddConstraint(k)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True, pruned=True)
for _ in range(5): table.launch()
table.eval(download=False, collect_only=True)
torch.cuda.synchronize()
for steps in (20, 20, 50, 100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): table.launch()
    t1 = time.perf_counter()
    r = table.eval(download=False, collect_only=True)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("steps %3d: total %.1f us = %.2f us/step | enqueue %.1f us (%.2f per launch) | collecting call %.1f us | sync %.1f us | kernel avg %.2f us" % (
        steps, (t3 - t0) * 1e6, (t3 - t0) * 1e6 / steps, (t1 - t0) * 1e6, (t1 - t0) * 1e6 / steps, (t2 - t1) * 1e6, (t3 - t2) * 1e6, r.fast_kernel_ms * 1e3))
# the collecting call alone, the device long idle: pure host-side cost of collection
for _ in range(3):
    for _ in range(20): table.launch()
    time.sleep(0.01)
    t1 = time.perf_counter()
    r = table.eval(download=False, collect_only=True)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("device idle: collecting call %.1f us | torch sync %.1f us" % ((t2 - t1) * 1e6, (t3 - t2) * 1e6))
# one launch from idle to collected: launch latency + one step + completion
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    table.launch()
    r = table.eval(download=False, collect_only=True)
    t1 = time.perf_counter()
    print("one step from idle: %.1f us (device step %.1f us)" % ((t1 - t0) * 1e6, r.fast_kernel_ms * 1e3))
