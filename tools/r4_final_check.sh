#!/bin/bash
# what the driver runs at round end, on the final tree: GPU suite, smoke, default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/r4_final_tests.log 2>&1
tail -3 $O/r4_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r4_final_smoke.log 2>&1
tail -4 $O/r4_final_smoke.log
python bench.py > $O/r4_final_bench.json 2> $O/r4_final_bench.err
python -c "
import json; d=json.loads(open('$O/r4_final_bench.json').read().strip().splitlines()[-1])
print(round(d['value'],1), 'images/s', round(d['ms_per_step'],2), 'ms', d['roofline']['name'], round(d['roofline']['frac'],3), 'cpu', round(d['cpu_baseline']['value'],3), 'ref', round(d['reference_mode_step']['images_per_sec'],1), 'f32', round(d['f32_step']['images_per_sec'],1))"
