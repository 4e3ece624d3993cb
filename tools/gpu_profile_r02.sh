#!/usr/bin/env bash
# Round 2: one GPU-box session that produces what profiles/r02_* keeps.  Every rocprofv3 pass is a run of its own (PMC
# passes are never combined with tracing), every step has its own time limit, and the script stops at the first failure.
# Usage (on the GPU box, from the repo root): bash tools/gpu_profile_r02.sh [part]    part: bench | k3 | k3extra | tree | all
set -uo pipefail
PART=${1:-all}
O=gpurun_out/prof_r02
mkdir -p $O
export TMPDIR=/tmp
run() { echo "== $*"; "$@"; rc=$?; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.1 --preroll 30"
if [ $PART = bench ] || [ $PART = all ]; then
  run timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- $B > $O/bench_stats.log 2>&1
  grep '"metric"' $O/bench_stats.log > $O/bench_line.json
  run timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench_wino -o bench -- $B --k3 wino > $O/bench_wino_stats.log 2>&1
  grep '"metric"' $O/bench_wino_stats.log > $O/bench_wino_line.json
  # HBM traffic of the bench's own K3 launches (PMC serialises the dispatches; counts are per launch all the same)
  run timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/bench_fetch -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.1 --preroll 8 > $O/bench_fetch.log 2>&1
  run timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/bench_write -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-leg --mean-plies 33.1 --preroll 8 > $O/bench_write.log 2>&1
fi
if [ $PART = k3 ] || [ $PART = all ]; then
  # the product K3 alone (tools/k3_sweep.py, 6144 leaves per launch = 2048 workgroups of 3 positions)
  run timeout -s KILL 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $O/k3_mfma -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_mfma.log 2>&1
  run timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/k3_insts -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_insts.log 2>&1
  run timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $O/k3_fetch -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_fetch.log 2>&1
  run timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $O/k3_write -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_write.log 2>&1
  # un-profiled timings: the kernel configurations side by side, the operand-removal builds, the in-kernel stamps
  for c in 34 44 24 48; do echo "== cfg $c" >> $O/k3_cfgs.log; AZ_K3_SPLIT_CFG=$c timeout -k 10 100 python3 tools/k3_sweep.py 1024 6144 16384 2>&1 | grep "n=" >> $O/k3_cfgs.log || exit 1; done
  for v in AZ_K3S_DBG_NOW AZ_K3S_DBG_NOB AZ_K3S_DBG_NOW+AZ_K3S_DBG_NOB AZ_K3S_DBG_LOADONLY; do echo "== $v" >> $O/k3_variants.log; AZAMD_LIB=dbgbuild/libazamd_$v.so timeout -k 10 100 python3 tools/k3_sweep.py 6144 2>&1 | grep "n=" >> $O/k3_variants.log || exit 1; done
  AZAMD_LIB=dbgbuild/libazamd_AZ_K3S_DBG_TS.so timeout -k 10 100 python3 tools/k3s_stamps.py 6144 > $O/k3_stamps.log 2>&1 || exit 1
  timeout -k 5 60 ./dbgbuild/coexec > $O/coexec.log 2>&1 || exit 1
  timeout -k 5 60 ./dbgbuild/fillers > $O/fillers.log 2>&1 || exit 1
  timeout -k 5 60 ./dbgbuild/valu_rate > $O/valu_rate.log 2>&1 || exit 1
  timeout -k 5 120 ./dbgbuild/mfma_power > $O/mfma_power.log 2>&1 || exit 1
fi
if [ $PART = k3 ] || [ $PART = all ] || [ $PART = k3extra ]; then
  # the same stamps on the operand-removal builds: equal cycles per workgroup, shorter launches = a higher clock
  for v in AZ_K3S_DBG_NOW+AZ_K3S_DBG_TS AZ_K3S_DBG_NOB+AZ_K3S_DBG_TS; do echo "== $v" >> $O/k3_stamps_variants.log; AZAMD_LIB=dbgbuild/libazamd_$v.so timeout -k 10 100 python3 tools/k3s_stamps.py 6144 2>&1 | grep "total\|entry" >> $O/k3_stamps_variants.log || exit 1; AZAMD_LIB=dbgbuild/libazamd_$v.so timeout -k 10 100 python3 tools/k3_sweep.py 6144 2>&1 | grep "n=" >> $O/k3_stamps_variants.log || exit 1; done
fi
if [ $PART = tree ] || [ $PART = all ]; then
  T="python3 tools/bench_configs.py k1 ttt nbttt pure unpack"
  run timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $O/tree -o tree -- $T > $O/tree_stats.log 2>&1
  run timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $O/tree_fetch -o pmc -- $T > $O/tree_fetch.log 2>&1
  run timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $O/tree_write -o pmc -- $T > $O/tree_write.log 2>&1
  run timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/tree_sq -o pmc -- $T > $O/tree_sq.log 2>&1
  for w in 4 2 1; do echo "== AZ_TREE_GPW=$w" >> $O/tree_gpw.log; AZ_TREE_GPW=$w timeout -k 10 200 python3 tools/bench_configs.py ttt nbttt 2>/dev/null >> $O/tree_gpw.log || exit 1; done
fi
echo done
