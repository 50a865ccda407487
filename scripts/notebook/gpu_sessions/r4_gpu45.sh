#!/bin/bash
# round 4, session 45 (PRODUCT build): what the thin maps (instance mask, colours) riding along with the wide map cost
set -u
export TMPDIR=/tmp
timeout -k 5 600 python scripts/exp_thin_share.py 2>&1 | grep -v amdgpu
