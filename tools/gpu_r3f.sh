#!/bin/bash
# multi-GPU evidence on the 1-GPU box: (a) --gpus 2 must fail loudly, (b) the RCCL path at world_size 1 with the configs[3] preset
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3f_gpus2.out 2> gpurun_out/r3f_gpus2.err; echo "rc=$?" >> gpurun_out/r3f_gpus2.err
tail -2 gpurun_out/r3f_gpus2.err
RVCMI_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 3 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3f_config3_rccl_world1.json 2> gpurun_out/r3f_config3.err
tail -3 gpurun_out/r3f_config3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3f_config3_rccl_world1.json'))
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config'])
print(d.get('repeats',{}).get('rank_median_ms_min_max'))
PY
RVCMI_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3f_config1_rccl_world1.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3f_config1_rccl_world1.json'))
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['index_broadcast_s'], d['config']['index_blob_bytes'], d['config']['ranks_first_search_equal'])
PY
