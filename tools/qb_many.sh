#!/bin/bash
# tools/quick_bench.sh for several library builds on one box, alternating: tools/qb_many.sh <rounds> <lib1> <lib2> ... (names as in ab_many.sh)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for L in "$@"; do
    case "$L" in
      shipped) P="";;
      exp) P="exp";;
      */*) P="$L";;
      *) P="$PWD/infinite-video_amd/libinfv_ltm_v_$L.so";;
    esac
    INFV_LTM_LIBRARY=$P tools/quick_bench.sh ${L}_$r 8 2>&1 | tail -1
  done
done
