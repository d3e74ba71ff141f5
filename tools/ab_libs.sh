#!/bin/bash
# A/B of two builds of libopenglottal_hip.so on the SAME box: tools/ab_libs.sh <libA> <libB> [chunk]
# (build the other variant with: git stash / checkout, csrc/build.sh, cp ../libopenglottal_hip.so ../libopenglottal_hip_prev.so)
A=$1; B=$2; C=${3:-64}
for r in 1 2; do for L in "$A" "$B"; do echo "== $(basename $L)"; OPENGLOTTAL_HIP_LIB=$L timeout -k 10 200 python tools/ab_option.py $C prio_mode 2 2>&1 | grep -E "median|chain"; done; done
