#!/bin/bash
# Same-box A/B of two builds of libkws_mi355x.so (box-to-box spread of the pool is larger than most single optimisations).
#   1. build the baseline:   git stash; make -C ei-keyword-spotting_amd/csrc OUT=$PWD/ab_tmp/libkws_old.so; git stash pop
#   2. build the candidate:  make -C ei-keyword-spotting_amd/csrc; cp ei-keyword-spotting_amd/libkws_mi355x.so ab_tmp/libkws_new.so
#   3. gpurun -- 'bash tools/ab_bench.sh'        (ab_tmp/ is scratch: git-ignored, but it travels to the GPU box)
#      (or 'bash tools/ab_bench.sh a b c': variants ab_tmp/libkws_{a,b,c}.so, the LAST one is the candidate that gets tested)
# Alternates the variants twice on the bench, then runs the GPU tests on the candidate.
V="${@:-old new}"; for c in $V; do LAST=$c; done
for v in $V $V; do
  cp ab_tmp/libkws_$v.so ei-keyword-spotting_amd/libkws_mi355x.so
  echo -n "$v: "; python bench.py --no-cpu-baseline | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], [x['kernel_ms'] for x in d['also']])"
done
cp ab_tmp/libkws_$LAST.so ei-keyword-spotting_amd/libkws_mi355x.so
[ -n "$AB_NO_TESTS" ] || timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | head -3
