set -x
python -m pytest tests/test_gpu_generator.py -q -k "streaming or full_clip or batch_16" 2>&1 | tail -30
python tools/diag_har.py 1198 > gpurun_out/diag_har.txt 2>&1; tail -20 gpurun_out/diag_har.txt
for mode in 0 auto small; do
  if [ $mode = 0 ]; then export RVCMI_RB_STREAM=0; unset RVCMI_RS_SMALL; fi
  if [ $mode = auto ]; then unset RVCMI_RB_STREAM; unset RVCMI_RS_SMALL; fi
  if [ $mode = small ]; then unset RVCMI_RB_STREAM; export RVCMI_RS_SMALL=1; fi
  python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/bench_r2b_$mode.json 2> gpurun_out/bench_r2b_$mode.err
  python -c "
import json;d=json.load(open('gpurun_out/bench_r2b_$mode.json'));r=d['roofline'];print('$mode',d['ms_per_step'],d.get('repeats',{}).get('ms_per_step_median'),r['kernel'],round(r['frac'],3),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
done
unset RVCMI_RS_SMALL RVCMI_RB_STREAM
python bench.py --no-cpu-baseline --repeats 2 --batch 16 --steps 10 --warmup 2 > gpurun_out/bench_r2b_b16.json 2> gpurun_out/bench_r2b_b16.err
python -c "
import json;d=json.load(open('gpurun_out/bench_r2b_b16.json'));r=d['roofline'];print('b16',d['ms_per_step'],d['value'],r['kernel'],round(r['frac'],3),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
RVCMI_RB_STREAM=0 python bench.py --no-cpu-baseline --repeats 2 --batch 16 --steps 10 --warmup 2 > gpurun_out/bench_r2b_b16_old.json 2> gpurun_out/bench_r2b_b16_old.err
python -c "
import json;d=json.load(open('gpurun_out/bench_r2b_b16_old.json'));r=d['roofline'];print('b16old',d['ms_per_step'],d['value'],r['kernel'],round(r['frac'],3),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
