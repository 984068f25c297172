#!/bin/bash
# HBM-side traffic of the walk kernel at the bench's default workload: two --pmc passes
# (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), one launch each.
OUT=gpurun_out/${1:-traffic}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 1 --warmup 0 --no-cpu"
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1 && echo fetch ok
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1 && echo write ok
timeout -s KILL 180 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/ea -- python3 bench.py $ARGS > $OUT/ea.log 2>&1 && echo ea ok
