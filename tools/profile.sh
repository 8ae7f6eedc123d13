#!/bin/bash
# rocprofv3 passes for the bench step (run on the GPU box).  Usage: tools/profile.sh <tag> [bench args...]
# Pass 1: kernel trace + stats.  Pass 2/3: PMC counters (own runs, no trace domains besides kernel).
set -u
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/prof_$TAG
KEEP=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $KEEP
ARGS="--steps 5 --warmup 2 --repeats 0 --no-cpu-baseline --no-extra --no-roofline --graph 0 $*"
# (the trace pass runs 30 + 8 steps: with 5 + 2 the cold first launches weighed 5 % on the per-kernel averages that bench.py's event timing is compared with)
TARGS="--steps 30 --warmup 8 --repeats 0 --no-cpu-baseline --no-extra --no-roofline --graph 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/bench.py $TARGS > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc1 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc3 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc3.log 2>&1
python $REPO/tools/summarize_prof.py $OUT > $KEEP/summary.txt 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $KEEP/kernel_stats.csv \;
cp $OUT/pmc_traffic.json $KEEP/ 2>/dev/null
tail -3 $OUT/trace.log > $KEEP/trace_tail.log
cat $KEEP/summary.txt
