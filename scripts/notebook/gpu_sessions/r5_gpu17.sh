#!/bin/bash
# round 5, session 17: microbenchmark -- do exec-masked ds_read_b128 (whole 16-lane groups off) cost fewer LDS cycles?
./build_ab/lds_partial_exec
