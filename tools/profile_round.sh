#!/bin/bash
# One profiling pass of the headline sweep (run on the GPU box from the repo root):  bash tools/profile_round.sh <tag>
#   <tag>/trace         rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu`  (+ trace_bench.json)
#   <tag>/pmc, mem      counter passes of the production kernel at 4e6 photons per atmosphere
#   <tag>/pmc32, mem32  the same for the opt-in one-block random stream (--rng philox32x4)
#   <tag>/pmc40, mem40  ... and for --rng philox40
#   <tag>/traffic       FETCH_SIZE / WRITE_SIZE at the full workload
#   <tag>/tau.log, variants.log
# Every rocprofv3 run has the program itself after `--` and a time limit of its own; a failed step ends the script.
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu > $OUT/trace_bench.json 2> $OUT/trace.err || { echo trace failed; tail -5 $OUT/trace.err; exit 1; }
echo trace ok
bash tools/profile_pmc.sh $TAG/pmc || exit 1
bash tools/pmc_mem.sh $TAG/mem || exit 1
bash tools/profile_pmc.sh $TAG/pmc32 --photons 4000000 --steps 1 --warmup 0 --no-cpu --rng philox32x4 || exit 1
PMC_MEM_ARGS="--rng philox32x4" bash tools/pmc_mem.sh $TAG/mem32 || exit 1
bash tools/profile_pmc.sh $TAG/pmc40 --photons 4000000 --steps 1 --warmup 0 --no-cpu --rng philox40 || exit 1
PMC_MEM_ARGS="--rng philox40" bash tools/pmc_mem.sh $TAG/mem40 || exit 1
bash tools/pmc_traffic.sh $TAG/traffic || exit 1
timeout -k 10 200 python tools/time_tau.py > $OUT/tau.log 2>/dev/null || exit 1
timeout -k 10 400 python tools/time_variants.py > $OUT/variants.log 2>/dev/null || exit 1
timeout -k 10 300 python tools/time_ranks.py > $OUT/ranks.log 2>/dev/null || exit 1
echo profile pass complete
