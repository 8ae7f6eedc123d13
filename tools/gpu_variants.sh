#!/bin/bash
# Several environment variants of ONE build on ONE gpurun lease, interleaved (boxes of the pool differ by 3-15 %): tools/gpu_variants.sh NAME ROUNDS "ENV_0" "ENV_1" ...
# ("" = the defaults).  Prints per run: ms_per_step (K timed steps), the repeat median and the per-kernel event times of the eager profiling pass.
name=$1; rounds=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  i=0
  for e in "$@"; do
    env $e python bench.py --no-cpu-baseline --no-gpu-torch-baseline --no-extra --repeats 3 ${BENCH_ARGS} > gpurun_out/var_${name}_${i}_${r}.json 2> gpurun_out/var_${name}_${i}_${r}.err
    python - "$name" $i $r "$e" <<'PY'
import json, sys
name, i, r, e = sys.argv[1:5]
try:
    d = json.load(open("gpurun_out/var_%s_%s_%s.json" % (name, i, r)))
except Exception as ex:
    print(name, i, r, "FAILED", ex); sys.exit(0)
rf = d.get("roofline", {})
k = rf.get("kernels_ms_per_step", {})
print("%s v%s r%s [%s] step %.4f median %s | %s" % (name, i, r, e, d["ms_per_step"], d.get("repeats", {}).get("ms_per_step_median"),
      " ".join("%s %.4f" % (n, v) for n, v in k.items() if v >= 0.03)))
PY
    i=$((i+1))
  done
done
