#!/usr/bin/env bash
# Round 3: one GPU-box session that produces what profiles/r03_* keeps.  Every rocprofv3 pass is a run of its own (PMC
# passes are never combined with tracing), every step has its own time limit, and the script stops at the first failure.
# Usage (on the GPU box, from the repo root): bash tools/gpu_profile_r03.sh [part]    part: bench | cfg3 | cfg4 | tree | shape | all
set -uo pipefail
PART=${1:-all}
O=gpurun_out/prof_r03
mkdir -p $O
export TMPDIR=/tmp
run() { echo "== $*"; "$@"; rc=$?; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.1 --preroll 12"
if [ $PART = bench ] || [ $PART = all ]; then
  run timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- $B > $O/bench_stats.log 2>&1
  grep '"metric"' $O/bench_stats.log > $O/bench_line.json
  run timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench_off -o bench -- $B --eval-cache -1 > $O/bench_off_stats.log 2>&1
  grep '"metric"' $O/bench_off_stats.log > $O/bench_off_line.json
  # the same without the HIP events of bench.py's own kernel timing: what the event packets add to the gaps between kernels
  run timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench_noev -o bench -- $B --no-profile > $O/bench_noev_stats.log 2>&1
  grep '"metric"' $O/bench_noev_stats.log > $O/bench_noev_line.json
  # HBM traffic of the bench's own K3 / tree-step launches (PMC serialises the dispatches; counts are per launch all the same)
  P="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.1 --preroll 4"
  run timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/bench_fetch -o pmc -- $P > $O/bench_fetch.log 2>&1
  run timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/bench_write -o pmc -- $P > $O/bench_write.log 2>&1
  run timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/bench_sq -o pmc -- $P > $O/bench_sq.log 2>&1
fi
if [ $PART = cfg3 ] || [ $PART = all ]; then
  C3="python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.2 --preroll 10"
  run timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/cfg3 -o bench -- $C3 > $O/cfg3_stats.log 2>&1
  grep '"metric"' $O/cfg3_stats.log > $O/cfg3_line.json
fi
if [ $PART = cfg4 ] || [ $PART = all ]; then
  C="python3 bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --mean-plies 41.1 --preroll 48"
  run timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/cfg4 -o bench -- $C > $O/cfg4_stats.log 2>&1
  grep '"metric"' $O/cfg4_stats.log > $O/cfg4_line.json
  run timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/cfg4_fetch -o pmc -- $C > $O/cfg4_fetch.log 2>&1
  run timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/cfg4_write -o pmc -- $C > $O/cfg4_write.log 2>&1
  run timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/cfg4_sq -o pmc -- $C > $O/cfg4_sq.log 2>&1
fi
if [ $PART = tree ] || [ $PART = all ]; then
  # games per wave of the fused / pure searches in the saturated regime: launch times un-profiled, then the SQ counters per row
  run timeout -k 10 420 bash tools/gpw_sweep.sh > $O/tree_gpw.log 2>&1
  for w in 4 2; do
    T="python3 tools/bench_configs.py ttt nbttt --ttt-games 16384 --nbttt-games 16384 --mean-len 20"
    AZ_TREE_GPW=$w run timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/tree_sq_gpw$w -o pmc -- $T > $O/tree_sq_gpw$w.log 2>&1
  done
  for w in 8 4; do
    T="python3 tools/bench_configs.py pure --pure-games 16384 --no-cpu"
    AZ_TREE_GPW=$w run timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/pure_sq_gpw$w -o pmc -- $T > $O/pure_sq_gpw$w.log 2>&1
  done
  T="python3 tools/bench_configs.py k1 ttt nbttt pure unpack --ttt-games 16384 --nbttt-games 16384 --pure-games 16384 --no-cpu"
  run timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $O/tree -o tree -- $T > $O/tree_stats.log 2>&1
  run timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $O/tree_fetch -o pmc -- $T > $O/tree_fetch.log 2>&1
  run timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $O/tree_write -o pmc -- $T > $O/tree_write.log 2>&1
fi
if [ $PART = shape ] || [ $PART = all ]; then
  [ -x gpurun_ab/mfma_shape ] && (timeout -k 5 120 ./gpurun_ab/mfma_shape > $O/mfma_shape.log 2>&1 || exit 1)
  AZ_TREE_STEP_GPW=8 AZ_ROLL_CAP=6 timeout -k 10 120 python3 tools/tree_step_alone.py 2048 4 > $O/tree_alone.log 2>&1 || exit 1
  AZ_TREE_STEP_GPW=2 AZ_ROLL_CAP=6 timeout -k 10 120 python3 tools/tree_step_alone.py 2048 4 >> $O/tree_alone.log 2>&1 || exit 1
fi
echo done
