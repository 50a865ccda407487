#!/bin/bash
# round 5, session 22 (EXPERIMENTS build): is the point order a permutation and the exact Hilbert order, with XCD-private counter planes and with one table?
export TMPDIR=/tmp D3F_BUILD_EXPERIMENTS=1
timeout -k 5 600 python scripts/notebook/exp_order_check.py 2>&1 | grep -v amdgpu | tail -12
