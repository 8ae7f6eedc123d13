#!/bin/bash
# Interleaved A/B of one environment switch on ONE gpurun lease (boxes of the pool differ by 3-5 %: an A/B across leases means
# nothing).  usage: tools/gpu_ab.sh NAME "ENV_A" "ENV_B" [rounds]   e.g.  tools/gpu_ab.sh y16 "RVCMI_Y_F16=0" "RVCMI_Y_F16=1" 3
# Prints per run: ms_per_step (K timed steps), the repeat median, and the per-kernel event times of the eager profiling pass.
name=$1; ea=$2; eb=$3; rounds=${4:-3}
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then e="$ea"; else e="$eb"; fi
    env $e python bench.py --no-cpu-baseline --no-gpu-torch-baseline --no-extra --repeats 5 > gpurun_out/ab_${name}_${v}${r}.json 2> gpurun_out/ab_${name}_${v}${r}.err
    python - "$name" $v $r "$e" <<'PY'
import json, sys
name, v, r, e = sys.argv[1:5]
try:
    d = json.load(open("gpurun_out/ab_%s_%s%s.json" % (name, v, r)))
except Exception as ex:
    print(name, v, r, "FAILED", ex); sys.exit(0)
k = {s["name"]: round(s["ms"], 4) for s in d.get("kernels", [])} if "kernels" in d else {}
rf = d.get("roofline", {})
print("%s %s%s [%s] ms_per_step %.4f median %s frac %.4f kernels %s" % (name, v, r, e, d["ms_per_step"], d.get("repeats", {}).get("ms_per_step_median"),
      rf.get("frac", 0), json.dumps(rf.get("kernels_ms_per_step", k))))
PY
  done
done
