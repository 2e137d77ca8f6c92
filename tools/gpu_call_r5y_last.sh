# round 5: the LAST of ~30 one-off GPU calls made while bisecting the nested-run issues (profiles/r05y_*.txt describe what each found); kept as the template of a short call
OUT=gpurun_out/r05z6; mkdir -p $OUT
timeout 30 python tools/dbg_nest.py 100345 102465 100036 2>&1 | grep "^10" | cut -c1-60 | tee $OUT/seeds.txt
timeout 60 python bench.py --steps 5 --warmup 1 --cpu-budget 0 --no-other 2>/dev/null | tail -n 1 > $OUT/headline.json; python -c "
import json; o = json.load(open('$OUT/headline.json')); print('headline', o['ms_per_step'], o['roofline']['kernel_avg_ms'], o['roofline']['kernel_isa_hash'])"
