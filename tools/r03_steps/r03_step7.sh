#!/bin/bash
# round 3, step 7: out-of-line transcendentals in the post-solve; why is config 1 (real femur) 3x the synthetic femur-size time?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_s7; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
V=("r02=libgingr_hip_r02.so" "cur=")
bash tools/abn.sh "${V[@]}" -- > $O/ab50k.txt 2>&1; cat $O/ab50k.txt
bash tools/abn.sh "${V[@]}" -- --points 1622 --steps 300 --warmup 10 > $O/ab1622.txt 2>&1; cat $O/ab1622.txt
bash tools/prof_emu8.sh > $O/prof_emu8.txt 2>&1; head -22 $O/prof_emu8.txt | cut -c1-150
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c1 --output-format csv -- python3 $R/bench.py --config 1 > $O/config1.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03_s7/prof_c1/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.3f}')
PY
tail -1 $O/config1.txt | cut -c1-300
