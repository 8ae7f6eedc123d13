#!/bin/bash
# One GPU session that refreshes every artefact of a round: tools/gpu_round.sh r06   (run through gpurun; ~15 min).
# Build tools/ubench/kloop2 first (hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o kloop2 kloop2.hip: the binary travels with the snapshot).
TAG=${1:-r06}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${TAG}_pytest_gpu_tail.txt; cat gpurun_out/${TAG}_pytest_gpu_tail.txt
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/${TAG}_bench_b1_T1198.json 2> gpurun_out/${TAG}_bench_b1.err; tail -1 gpurun_out/${TAG}_bench_b1.err
python bench.py --batch 64 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/${TAG}_bench_b64_T1198.json 2>/dev/null
python bench.py --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/${TAG}_bench_b16_T1198.json 2>/dev/null
python bench.py --whole --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/${TAG}_bench_whole_b16_T1198.json 2>/dev/null
python bench.py --stream > gpurun_out/${TAG}_bench_stream_v1_40k.json 2>/dev/null
python bench.py --e2e > gpurun_out/${TAG}_bench_e2e.json 2>gpurun_out/${TAG}_bench_e2e.err
python tools/gru_time.py > gpurun_out/${TAG}_gru_time.txt 2>/dev/null   # GRUHIP vs torch nn.GRU (MIOpen) at the realtime and clip window sizes
# the K loop's own ceiling on THIS chip, re-measured every round (bench.py parses the newest profiles/rNN_ubench_kloop2_issue_model.txt)
[ -x tools/ubench/kloop2 ] && (cd tools/ubench && timeout 300 ./kloop2) > gpurun_out/${TAG}_ubench_kloop2_issue_model.txt 2>&1
# the RCCL path at world size 1 (the one-GPU lease): configs[1] and configs[3] (64 clips, 1M x 256 index built on the GPU, ONE timed broadcast, agreement check)
RVCMI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --no-cpu-baseline --no-gpu-torch-baseline --no-extra > gpurun_out/${TAG}_bench_config1_rccl_world1.json 2>/dev/null
RVCMI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --config 3 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/${TAG}_bench_config3_rccl_world1.json 2>/dev/null
bash tools/profile.sh $TAG > gpurun_out/prof_${TAG}.log 2>&1; tail -1 gpurun_out/prof_${TAG}.log
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("bench_b1_T1198", "bench_b64_T1198", "bench_b16_T1198", "bench_whole_b16_T1198"):
    try:
        d = json.load(open("gpurun_out/%s_%s.json" % (tag, f))); r = d.get("roofline") or {}
        print(f, round(d["ms_per_step"], 4), round(d["value"], 1), d.get("repeats", {}).get("ms_per_step_median"), r.get("frac"), r.get("traffic"))
    except Exception as e:
        print(f, "FAILED", e)
try:
    d = json.load(open("gpurun_out/%s_bench_stream_v1_40k.json" % tag)); print("stream", d["hot_path"], d["whole_chunk"]["p50_ms"], d["whole_chunk"]["p99_ms"])
    d = json.load(open("gpurun_out/%s_bench_b1_T1198.json" % tag))
    print("whole", d["whole_infer"]["ms_per_step"], d["whole_infer"]["value"], "torch", d["gpu_torch_baseline"]["fp16"]["ms_per_clip"], d["gpu_torch_baseline"]["fp32"]["ms_per_clip"], "cpu", d["cpu_baseline"]["value"])
except Exception as e:
    print("FAILED", e)
PY
