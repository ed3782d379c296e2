#!/bin/bash
# alternating passes of two builds / settings on one box
one() { local label=$1; shift; echo -n "$label : "
  env "$@" timeout 300 python tools/one_pass.py ${CHUNKS:-2048} 6 2>&1 | grep "^pass" | tail -4 | awk '{print $3}' | sort -n | tr '\n' ' '; echo; }
for r in 1 2 3; do
one "r04 library        " INFV_LTM_LIBRARY=$PWD/infinite-video_amd/libinfv_ltm_r04exp.so
one "shipped            " INFV_LTM_LIBRARY=
done
CHUNKS=256 one "256: r04 library   " INFV_LTM_LIBRARY=$PWD/infinite-video_amd/libinfv_ltm_r04exp.so
CHUNKS=256 one "256: shipped       " INFV_LTM_LIBRARY=
