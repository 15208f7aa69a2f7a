#!/usr/bin/env python3
"""Long differential campaign of the policy compiler: tests/test_template_fuzz.py's generator over a range of seeds, on the
GPU-less test build (product through the C ABI vs the oracle).  Prints only what is not clean.

  python tools/fuzz_campaign.py FIRST LAST [--templates 150] [--objects 14] [--backend hostemu-gen] [--kernel-emu]

Seeds cycle through the generator's modes (plain / AdmissionRequest envelope / numeric edges / Rego v1 syntax).
--kernel-emu additionally runs every evaluation on the kernel emulator (GK_HOSTEMU_KERNEL=jit: the dominant kernel's HIP
source on fibres, compared bit for bit with the per-review evaluation) -- about 10x slower."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int)
    ap.add_argument("last", type=int)
    ap.add_argument("--templates", type=int, default=150)
    ap.add_argument("--objects", type=int, default=14)
    ap.add_argument("--backend", default="hostemu-gen", choices=["hostemu", "hostemu-gen"])
    ap.add_argument("--kernel-emu", action="store_true")
    a = ap.parse_args()
    if a.kernel_emu:
        os.environ["GK_HOSTEMU_KERNEL"] = "jit"
        os.environ.setdefault("GK_EMU_GRID", "8")
    import test_template_fuzz as F
    total = {}
    for seed in range(a.first, a.last + 1):
        mode = seed % 4
        stats, diffs = F.run(a.backend, seed, a.templates, a.objects, envelope=mode == 1, numeric=mode >= 2, v1=mode == 3)
        for k, v in stats.items():
            total[k] = total.get(k, 0) + v
        if diffs or stats["oracle_err"] or stats["product_err"]:
            print("=== seed %d (mode %d) %s" % (seed, mode, stats))
            for bad, rego, objs in diffs[:3]:
                print(bad)
                print(rego)
                print(objs)
            sys.stdout.flush()
    print("seeds %d..%d: %s" % (a.first, a.last, total))


if __name__ == "__main__":
    main()
