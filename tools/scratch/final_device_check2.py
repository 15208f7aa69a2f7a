#!/usr/bin/env python3
"""The product-against-compiled-checker tests of tests/test_indep_check.py run ON THE DEVICE (they are CPU-build tests in the suite):
every HandleReview shape (20 000 reviews x 52 constraints: bitmaps + statuses), RESULT totals, message text, bitmaps at 20 000 x 50 --
then, while the time budget lasts, the same on the 200-constraint corpus.  Torch-free.  GK_CHECK_ON_CPU_BUILD=1: dry run."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
BUDGET_S = float(os.environ.get("GK_CHECK_BUDGET_S", "75"))
t0 = time.time()
from gatekeeper_amd import driver as D   # noqa: E402
from gatekeeper_amd import synth          # noqa: E402

ON_CPU = os.environ.get("GK_CHECK_ON_CPU_BUILD") == "1"
_Driver = D.Driver


class DeviceDriver(_Driver):
    def __init__(self, device=0, hostemu=None, **kw):
        super().__init__(device=device, hostemu=ON_CPU, **kw)


D.Driver = DeviceDriver
import test_indep_check as T              # noqa: E402

fx = synth.load_fixtures()
steps = [("review_shapes_20000x52", lambda: T.test_review_shapes_product_equals_the_compiled_checker(fx)),
         ("result_totals_audit50_6000", lambda: T.test_result_totals_product_equals_the_compiled_checker("audit-50", 6000, fx)),
         ("messages_audit50_4000", lambda: T.test_product_messages_equal_the_compiled_checker("audit-50", 4000, fx)),
         ("bitmaps_audit50_20000", lambda: T.test_product_equals_the_compiled_checker_at_sizes_the_python_oracle_does_not_reach("audit-50", 20000, fx)),
         ("result_totals_corpus_2500", lambda: T.test_result_totals_product_equals_the_compiled_checker("corpus-200", 2500, fx)),
         ("bitmaps_corpus_3000", lambda: T.test_product_equals_the_compiled_checker_at_sizes_the_python_oracle_does_not_reach("corpus-200", 3000, fx)),
         ("messages_corpus_600", lambda: T.test_product_messages_equal_the_compiled_checker("corpus-200", 600, fx))]
only = [x for x in os.environ.get("GK_CHECK_STEPS", "").split(",") if x]
steps = [st for st in steps if not only or st[0] in only]
out = {"backend": "cpu build (dry run)" if ON_CPU else "device", "steps": {}}
for name, fn in steps:
    if time.time() - t0 > BUDGET_S:
        out["steps"][name] = "skipped (time budget)"
        continue
    t = time.time()
    try:
        fn()
        out["steps"][name] = {"equal": True, "seconds": round(time.time() - t, 1)}
    except AssertionError as ex:
        out["steps"][name] = {"equal": False, "what": str(ex)[:300]}
    except Exception as ex:   # noqa: BLE001
        out["steps"][name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
    print(json.dumps({name: out["steps"][name]}), flush=True)
out["seconds"] = round(time.time() - t0, 1)
print(json.dumps(out))
