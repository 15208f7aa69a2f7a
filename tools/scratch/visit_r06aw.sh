#!/bin/bash
# round 6, visit aw: the default bench line and the stream legs of the totals-rows build; the faster gk_sum_partials under rocprofv3
set -u
timeout 600 python -m pytest tests/test_parity.py tests/test_result_totals.py -m gpu -x -q 2>&1 | tail -2
bash tools/gpu_visit.sh r06aw bench stats stream
