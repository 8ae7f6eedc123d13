for nb in 2 3 4; do
RVCMI_RS_NB=$nb python bench.py --no-cpu-baseline --repeats 0 --steps 50 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('NB=$nb',round(d['ms_per_step'],4),r['kernels_ms_per_step']['rb_stream_c128'])"
done
for v in 0 1 2 3; do
RVCMI_IVF_VAR=$v python bench.py --no-cpu-baseline --repeats 0 --steps 50 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('IVF_VAR=$v',round(d['ms_per_step'],4),r['kernels_ms_per_step']['ivf_scan'],r['kernels_ms_per_step']['ivf_coarse'])"
done
python -m pytest tests/test_gpu_ivf.py -q -k "bit_exact or real_hubert" 2>&1 | tail -2
RVCMI_IVF_VAR=3 python -m pytest tests/test_gpu_ivf.py -q -k "bit_exact or real_hubert or blend" 2>&1 | tail -2
