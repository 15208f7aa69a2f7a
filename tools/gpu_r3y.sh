#!/bin/bash
# Round-3 visit y: the RCCL test variants after the overlapped passes leave workgroups to the exchange stream
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 100 python -m pytest tests/test_sweep_dist.py -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r3y_pytest_rccl.log
