# round 5: the LAST of ~35 short GPU calls made while bisecting the nested-run issues (profiles/r05y_*.txt describe what each found); kept as the template of a short call
OUT=gpurun_out/r05z7; mkdir -p $OUT
timeout 25 python -m pytest tests -q -m gpu -x -k "stream or gang or reused or handle" -p no:cacheprovider 2>&1 | tail -n 2 | tee $OUT/pytest_subset.txt
