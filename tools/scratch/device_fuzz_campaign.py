#!/usr/bin/env python3
"""Differential fuzz ON THE DEVICE: tests/test_template_fuzz.py's random templates, ten to a plan, each plan through hiprtc and the
plan-specialised kernel on the MI355X, compared with the oracle per template and review.
  python tools/scratch/device_fuzz_campaign.py FIRST LAST [--templates 60] [--objects 14]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("first", type=int); ap.add_argument("last", type=int)
ap.add_argument("--templates", type=int, default=60); ap.add_argument("--objects", type=int, default=14)
ap.add_argument("--backend", default="gpu", help="gpu, or hostemu: the same campaign on the GPU-less test build")
a = ap.parse_args()
import test_template_fuzz as F
loaded = compared = bad = 0
t0 = time.time()
for seed in range(a.first, a.last + 1):
    mode = seed % 4
    try:
        l, c = F.run_batched(a.backend, seed, a.templates, a.objects, envelope=mode == 1, numeric=mode >= 2, v1=mode == 3)
        loaded += l; compared += c
    except Exception as e:   # (an EngineError -- device and renderer disagree -- is a failing seed too, not the end of the campaign)
        bad += 1
        print("=== seed %d (mode %d): %s" % (seed, mode, str(e)[:2000])); sys.stdout.flush()
print("device fuzz seeds %d..%d: %d templates compiled and loaded, %d (review, plan) results compared, %d failing seeds, %.0f s" % (a.first, a.last, loaded, compared, bad, time.time() - t0))
