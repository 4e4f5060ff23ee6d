#!/bin/bash
# round 6: deep soaks on the final library -- every clip of 4 full batches per model against the oracle, exact mode (bit for bit) and fast mode (against its tolerance)
set -u
mkdir -p gpurun_out
(time timeout 1500 python tests/deep_soak.py 4 65536) > gpurun_out/r06_soak_exact.txt 2>&1; tail -12 gpurun_out/r06_soak_exact.txt
(time timeout 1500 python tests/deep_soak.py 4 65536 fast) > gpurun_out/r06_soak_fast.txt 2>&1; tail -12 gpurun_out/r06_soak_fast.txt
