#!/bin/bash
# One parametrised gpurun job (replaces the per-call step scripts of earlier rounds).  Runs, in this order and each only if asked for:
#   TESTS="<pytest -k expression or file list>"   GPU tests (TESTS=all: the whole -m gpu suite)
#   AB="<variant> <variant> ..."                  tools/abn.sh variants, e.g. "r03=tools/bin/libgingr_hip_r03.so cur="
#   WORKLOADS="50k emu8 15k 1622 ..."             bench workloads for the A/B and the kernel statistics (see wl_args below)
#   STATS="cur r03 ..."                           rocprofv3 kernel statistics per workload for these variants (lib as in AB)
#   EXTRA="<shell command>"                       anything else, run last
# usage: gpurun -- 'TAG=r04_s1 TESTS=all AB="r03=tools/bin/libgingr_hip_r03.so cur=" WORKLOADS="50k emu8" STATS="cur" bash tools/gpu_job.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/${TAG:-job}; mkdir -p $O
wl_args() {
  case $1 in
    50k)   echo "--steps 100 --warmup 10 --roofline-steps 0";;
    emu8)  echo "--emulate-world 8 --steps 100 --warmup 10 --roofline-steps 0";;
    emu4)  echo "--emulate-world 4 --steps 100 --warmup 10 --roofline-steps 0";;
    emu2)  echo "--emulate-world 2 --steps 100 --warmup 10 --roofline-steps 0";;
    15k)   echo "--points 15000 --steps 200 --warmup 10 --roofline-steps 0";;
    1622)  echo "--points 1622 --steps 300 --warmup 20 --roofline-steps 0";;
    4k)    echo "--points 4000 --steps 300 --warmup 20 --roofline-steps 0";;
    8k)    echo "--points 8000 --steps 300 --warmup 20 --roofline-steps 0";;
    100k)  echo "--points 100000 --steps 30 --warmup 5 --roofline-steps 0";;
    late)  echo "--sigma2 4 --steps 50 --warmup 5 --roofline-steps 0";;
    *)     echo "$1";;
  esac
}
if [ -n "$TESTS" ]; then
  if [ "$TESTS" = all ]; then timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
  else timeout 2400 python3 -m pytest tests -m gpu -x -q $TESTS > $O/pytest.txt 2>&1; fi
  echo "rc=$?" >> $O/pytest.txt; tail -15 $O/pytest.txt | cut -c1-220
fi
if [ -n "$AB" ]; then
  for w in ${WORKLOADS:-50k}; do
    bash tools/abn.sh $AB -- $(wl_args $w) > $O/ab_$w.txt 2>&1; echo "== A/B $w"; cut -c1-110 $O/ab_$w.txt
  done
fi
for v in $STATS; do
  name=${v%%=*}; lib=${v#*=}; [ "$lib" = "$v" ] && lib=""
  for w in ${WORKLOADS:-50k}; do
    ( if [ -n "$lib" ]; then export GINGR_HIP_LIB=$R/$lib GINGR_HIP_LIB_ALLOW_OLDER=1; fi
      bash tools/prof_stats.sh ${TAG:-job}_${name}_$w $(wl_args $w) > $O/stats_${name}_$w.txt 2>&1 )
    echo "== kernel stats $name $w"; head -22 $O/stats_${name}_$w.txt | cut -c1-150
  done
done
if [ -n "$EXTRA" ]; then eval "$EXTRA" 2>&1 | tail -40; fi
