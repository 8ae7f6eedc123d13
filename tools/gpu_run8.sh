python -m pytest tests -m gpu -q 2>&1 | tail -8
for sm in 1 0; do
RVCMI_RS_SMALL=$sm python bench.py --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('small=$sm',d['ms_per_step'],d['repeats']['ms_per_step_median'],r['kernel'],round(r['frac'],3),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
done
python bench.py --no-cpu-baseline --repeats 2 --batch 16 --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('b16',d['ms_per_step'],d['value'],r['kernel'],round(r['frac'],3),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
