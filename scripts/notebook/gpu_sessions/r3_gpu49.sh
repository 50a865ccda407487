#!/bin/bash
set -u
export D3F_BUILD_EXPERIMENTS=1
timeout -k 5 600 python -m pytest tests/test_gpu_walks.py -m gpu -q -x -k "window_gather_is_bit_identical" 2>&1 | tail -40 | cut -c1-220
