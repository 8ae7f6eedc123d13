#!/bin/bash
# Launcher penalty: the same 1-GPU step single-process vs under torch.distributed.run at world 1 (RCCL
# initialised, one broadcast of the index), interleaved on ONE lease.  Prints ms_per_step, repeat median and the dominant kernel's time.
mkdir -p gpurun_out
B="--no-cpu-baseline --no-gpu-torch-baseline --repeats 5"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["roofline"]["kernels_ms_per_step"]
    print("%-34s ms_per_step %.4f median %.4f rb_stream %.4f rb_full64 %.4f rb_pair %.4f launcher=%s" % (sys.argv[1], d["ms_per_step"], d["repeats"]["ms_per_step_median"],
          k.get("rb_stream_c128", 0), k.get("rb_full_c64", 0), k.get("rb_pair_c256", 0), d["config"].get("launcher")))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for r in 1 2; do
  python bench.py $B > gpurun_out/la_single$r.json 2> gpurun_out/la_single$r.err; show "single-process #$r" gpurun_out/la_single$r.json
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2950$r bench.py --gpus 1 $B > gpurun_out/la_torchrun$r.json 2> gpurun_out/la_torchrun$r.err
  show "torchrun world 1 (no RCCL) #$r" gpurun_out/la_torchrun$r.json
  RVCMI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$r bench.py --gpus 1 $B > gpurun_out/la_rccl$r.json 2> gpurun_out/la_rccl$r.err
  show "torchrun world 1 + RCCL #$r" gpurun_out/la_rccl$r.json
  RVCMI_BENCH_FORCE_DIST=1 OMP_NUM_THREADS=16 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$r bench.py --gpus 1 $B > gpurun_out/la_rccl_omp$r.json 2> gpurun_out/la_rccl_omp$r.err
  show "torchrun + RCCL, OMP_NUM_THREADS=16 #$r" gpurun_out/la_rccl_omp$r.json
  RVCMI_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$r RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --gpus 1 $B > gpurun_out/la_rccl_noltr$r.json 2> gpurun_out/la_rccl_noltr$r.err
  show "RCCL, env rendezvous, no torchrun #$r" gpurun_out/la_rccl_noltr$r.json
done
