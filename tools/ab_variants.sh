#!/bin/bash
# Same-session A/B of library builds on chosen kernels: bash tools/ab_variants.sh "a b" "mc,reflect,hg,depth;gray,reemit,hg" [photons]
VARS=$1; ONLY=$2; N=${3:-4000000}
for round in 1 2; do
for v in $VARS; do
  echo "== $v (round $round)"
  RT_AMD_LIB=$PWD/radiative-transfer_amd/csrc/variants/$v.so timeout -k 10 300 python tools/time_variants.py $N "$ONLY" 2>&1 | awk '{ i = index($0, "}"); print substr($0, i + 1) }' | sed 's/  */ /g' || exit 1
done; done
