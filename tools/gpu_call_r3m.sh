#!/usr/bin/env bash
# round 3, call m: pricer on the GPU + the LDS layout sweep (tools/lds_layout_sweep.sh built the variants)
OUT=gpurun_out/${1:-r03m}; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=120
timeout 900 python -m pytest tests/test_z_pricer.py tests/test_z_optimiser.py -q -m gpu > $OUT/pytest_pricer.log 2>&1; echo "pytest(pricer) rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_pricer.log | tee -a $OUT/summary.txt
bash tools/lds_layout_run.sh ${1:-r03m}
