#!/bin/bash
# Same-box A/B of two builds of the library: alternating passes of the headline call (tools/one_pass.py), best and median pass.
# usage: tools/ab_libs.sh <libA.so|exp|""> <libB.so|exp|""> [rounds] [chunks]
A=$1; B=$2; rounds=${3:-3}; chunks=${4:-2048}
for r in $(seq 1 $rounds); do
  for L in "$A" "$B"; do
    echo -n "lib=${L:-shipped} : "
    INFV_LTM_LIBRARY=$L python tools/one_pass.py $chunks 6 2>&1 | grep "^pass" | tail -4 | awk '{print $3}' | sort -n | tr '\n' ' '
    echo
  done
done
