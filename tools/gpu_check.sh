#!/bin/bash
# quick validation of a build on one lease: the whole GPU suite, smoke, and the default bench line (tools/gpu_round.sh refreshes every artefact)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; tail -2 gpurun_out/check_bench.err | cut -c1-200
python - <<'PY'
import json
d = json.load(open("gpurun_out/check_bench.json"))
print("step", round(d["ms_per_step"], 4), round(d["value"], 1), "frac", round(d["roofline"]["frac"], 4), "whole", round(d["whole_infer"]["ms_per_step"], 4), "front", d["whole_infer"]["front_ms_per_step"])
print("batch64", round(d["batch64"]["value"], 1), "stream", round(d["stream"]["hot_path"]["p50_ms"], 4), round(d["stream"]["whole_chunk"]["p50_ms"], 4), "bf16", round(d["bf16"]["bf16"]["value"], 1), d["bf16"]["bf16"]["rms_vs_fp32_kernels"])
print("ubench", d["roofline"].get("ubench_ceiling", {}).get("source"), "traffic", d["roofline"]["traffic_source"][:60] if d["roofline"].get("traffic_source") else None)
PY
