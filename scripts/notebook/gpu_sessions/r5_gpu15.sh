#!/bin/bash
# round 5, session 15 (PRODUCT build): rocprofv3 kernel-trace + PMC summaries of every workload (profiles/r5_v1)
set -u
export TMPDIR=/tmp
bash scripts/r5_profile_all.sh r5_v1 2>&1 | tail -20
