#!/usr/bin/env bash
# One GPU-box session that produces everything profiles/ keeps for a round: kernel-trace stats of bench.py and the
# PMC passes (each in a run of its own, never combined with tracing).  Usage: tools/gpu_profile_round.sh r01 [tree]
# (second argument "tree": only the tree / env / unpack passes)
set -uo pipefail
R=${1:-rXX}
O=gpurun_out/prof_$R
mkdir -p $O
export TMPDIR=/tmp
run() { echo "== $*"; "$@"; rc=$?; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
if [ "${2:-all}" != tree ]; then
run timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --mean-plies 33.27 > $O/bench_stats.log 2>&1
grep '"metric"' $O/bench_stats.log > $O/bench_line.json
# HBM traffic of the bench's own K3 launches (PMC serialises the dispatches; counts are per launch all the same)
run timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/bench_fetch -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --mean-plies 33.27 > $O/bench_fetch.log 2>&1
run timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/bench_write -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --mean-plies 33.27 > $O/bench_write.log 2>&1
# K3 alone (tools/k3_sweep.py, 6144 leaves per launch): MFMA utilisation, then HBM traffic
run timeout -s KILL 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $O/k3_mfma -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_mfma.log 2>&1
run timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $O/k3_fetch -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_fetch.log 2>&1
run timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $O/k3_write -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_write.log 2>&1
fi
# tree + env + unpack kernels (tools/bench_configs.py): HBM traffic
run timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $O/tree_fetch -o pmc -- python3 tools/bench_configs.py k1 ttt nbttt pure unpack > $O/tree_fetch.log 2>&1
run timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $O/tree_write -o pmc -- python3 tools/bench_configs.py k1 ttt nbttt pure unpack > $O/tree_write.log 2>&1
run timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $O/tree -o tree -- python3 tools/bench_configs.py k1 ttt nbttt pure unpack > $O/tree_stats.log 2>&1
echo done
