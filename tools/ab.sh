#!/bin/bash
# A/B timing of library builds in one GPU session: bash tools/ab.sh "A B" [bench args]
VARS=$1; shift
ARGS=${@:---steps 4 --warmup 1 --no-cpu}
for round in 1 2; do
for v in $VARS; do
  RT_AMD_LIB=$PWD/radiative-transfer_amd/csrc/variants/$v.so timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('$v round $round: %.3e events/s  step %.2f ms  kernel %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done; done
