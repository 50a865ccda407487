#!/bin/bash
set -u
timeout -k 5 600 python -m pytest tests -m gpu -q -x -k "track_run or rigid" 2>&1 | tail -15
