#!/bin/bash
# round 6, lease b: k_post_dma parity + ABAB, ups_c256 geometry variants, scale-invariance diagnostic
mkdir -p gpurun_out
python -m pytest tests/test_gpu_generator.py -m gpu -q -x 2>&1 | tail -5
python tools/diag_scale.py 200 > gpurun_out/r6b_diag_scale.txt 2>&1; cat gpurun_out/r6b_diag_scale.txt
bash tools/gpu_variants.sh post 2 "" "RVCMI_POST_DMA=0" "RVCMI_POST_DMA_OCC=2" "RVCMI_POST_DMA_OCC=4" "RVCMI_POST_DMA_OCC=6" 2>&1 | sed -e 's/noise_mfma_c256 [0-9.]* //' | cut -c1-420
bash tools/gpu_variants.sh ups256 2 "" "RVCMI_UPS_NJ_256=2" "RVCMI_UPS_NJ_256=4" "RVCMI_UPS_NJ_256=2 RVCMI_UPS_VPW_256=1" "RVCMI_UPS_NJ_256=4 RVCMI_UPS_VPW_256=1" "RVCMI_UPS_NJ_256=4 RVCMI_UPS_VPW_256=2" "RVCMI_UPS_NJ_256=1 RVCMI_UPS_VPW_256=2" "RVCMI_UPS_NJ_256=2 RVCMI_UPS_VPW_256=2" 2>&1 | cut -c1-420
