#!/bin/bash
# round 3, step 33: config lines again (config 2 now also reports the search a registration runs on the same pair)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s33; mkdir -p $O; cd $R
timeout 1500 python3 tools/bench_configs.py > $O/configs.txt 2> $O/configs.err; cp gpurun_out/r03_configs.json $O/configs.json; tail -6 $O/configs.txt | cut -c1-200
python3 -c "
import json; d=json.load(open('$O/configs.json'))
for x in d: print(x['config'], round(x['value'],1), round(x['ms_per_step'],4), x.get('registration_path'))"
