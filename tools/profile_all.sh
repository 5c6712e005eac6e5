#!/bin/bash
# Runs on the GPU box: one profile round (kernel trace + PMC passes) per workload of bench.py.
# Usage: bash tools/profile_all.sh <round-tag>     e.g. r02e  ->  gpurun_out/prof_r02e_<workload>/
TAG=${1:-r02}
LIST=${PROFILE_WORKLOADS:-"cornell-box veach-mis bathroom2 cornell-ct s0-rays-cornell s0-rays-cornell-coherent s4-rays-soup8m cornell-box-f32 veach-mis-f32 bathroom2-f32"}
for w in $LIST; do
  extra=""
  case "$w" in veach-mis*) extra="--spp 600";; esac     # 5 s of kernel per pass is plenty; counters scale with rays
  bash tools/profile_round.sh ${TAG}_$w --workload $w $extra > gpurun_out/p_${TAG}_$w.log 2>&1 || echo "$w failed"
done
ls gpurun_out | grep prof_${TAG}
