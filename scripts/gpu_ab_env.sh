#!/bin/bash
# A/B of environment switches on ONE library: AB_ENV="name:VAR=v[,VAR2=v2] ..." bash scripts/gpu_ab_env.sh TAG   (fitness words of every
# variant compared bit for bit with the default run; headline timings)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}
cd $R
bash scripts/gpu_div_ab.sh $TAG | grep "headline call\|IDENTICAL\|DIFFERENT\|cmp new"
