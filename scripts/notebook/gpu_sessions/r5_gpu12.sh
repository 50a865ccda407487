#!/bin/bash
# round 5, session 12 (PRODUCT build): SQ counters of pairwise_dist_kernel (the c5 step's 68 %): where do 213 M vs 180 M VALU instructions and the other 39 % of the issue slots go?
set -u
export TMPDIR=/tmp
python -m d3fields_amd.build > /dev/null 2>&1
bash scripts/notebook/pmc_any.sh r5_pairwise pairwise_dist_kernel python $(pwd)/scripts/notebook/run_pairwise.py 2>&1 | tail -30
