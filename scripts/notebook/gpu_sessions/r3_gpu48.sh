#!/bin/bash
# round 3, session 48: the GPU suite on the EXPERIMENTS build (the seven variant tests the product build skips)
set -u
export D3F_BUILD_EXPERIMENTS=1
timeout -k 5 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
