#!/bin/bash
# Same-box A/B of several library builds: alternating runs of the headline call (tools/one_pass.py), sorted pass times per run.
# usage: tools/ab_many.sh <rounds> <chunks> <lib1> <lib2> ...   (lib = "shipped" | "exp" | variant name of tools/build_variant.sh | path)
rounds=$1; chunks=$2; shift 2
for r in $(seq 1 $rounds); do
  for L in "$@"; do
    case "$L" in
      shipped) P="";;
      exp) P="exp";;
      */*) P="$L";;
      *) P="$PWD/infinite-video_amd/libinfv_ltm_v_$L.so";;
    esac
    echo -n "lib=$L : "
    INFV_LTM_LIBRARY=$P timeout 600 python tools/one_pass.py $chunks 7 2>&1 | grep "^pass" | tail -5 | awk '{print $3}' | sort -n | tr '\n' ' '
    echo
  done
done
