#!/bin/bash
# rocprofv3 kernel trace of the bench step (run on the GPU box): tools/prof_trace.sh <tag> [ENV=VAL ...] -- [bench args]
# Prints the per-kernel table (calls, total, average) and keeps the stats csv under gpurun_out/prof_<tag>/.
TAG=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/proft_$TAG
KEEP=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $KEEP
env "${ENVS[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/bench.py --steps 20 --warmup 2 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --no-roofline --graph 0 "$@" > $OUT/trace.log 2>&1
python $REPO/tools/summarize_prof.py $OUT > $KEEP/summary.txt 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $KEEP/kernel_stats.csv \;
tail -2 $OUT/trace.log > $KEEP/trace_tail.log
cat $KEEP/summary.txt
