#!/bin/bash
# single-latent steady-state step and shipped rollout in both math modes
R=$GRAFT_REPO_ROOT
for m in fp32 bf16x3; do
  SF_MATH_MODE=$m python3 $R/tools/chainbench.py euler 2>&1 | grep chain
  SF_MATH_MODE=$m python3 $R/tools/r02/rolltime.py shipped euler 2>&1 | grep rollout
  SF_MATH_MODE=$m python3 $R/tools/r02/rolltime.py stream40 euler 2>&1 | grep rollout
done
