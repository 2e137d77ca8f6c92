OUT=gpurun_out/r05y29; mkdir -p $OUT
timeout 400 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log
SOAK_LIB=hip SOAK_PROGRESS=1 timeout 100 python tests/soak.py rounds 2000 > $OUT/rounds_progress.txt 2>&1; echo "rounds rc=$? (124 = the time limit, not a hang, when the last seed is far beyond 100036)"; grep -v "^seed [0-9]*$" $OUT/rounds_progress.txt | grep -v amdgpu | tail -n 3; tail -n 1 $OUT/rounds_progress.txt
SOAK_LIB=hip timeout 80 python tests/soak.py streams 400 2>&1 | tail -n 1 | tee $OUT/streams.txt
