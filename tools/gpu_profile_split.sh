#!/usr/bin/env bash
# rocprofv3 passes for the experimental split-fp16 K3 (az_cnn_split.hip): kernel-trace stats of bench.py --k3 split and
# PMC passes on the kernel alone (each counter set in a run of its own, never combined with tracing).
# Usage: tools/gpu_profile_split.sh r01    (then: python tools/make_profiles.py r01)
set -uo pipefail
R=${1:-rXX}
O=gpurun_out/prof_${R}_split
mkdir -p $O
export TMPDIR=/tmp
run() { echo "== $*"; "$@"; rc=$?; echo "rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
run timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python3 bench.py --k3 split --steps 4 --warmup 1 --no-cpu-baseline --mean-plies 33.27 > $O/bench_stats.log 2>&1
grep '"metric"' $O/bench_stats.log > $O/bench_line.json
export AZ_K3_SPLIT=1
run timeout -s KILL 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $O/k3_mfma -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_mfma.log 2>&1
run timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $O/k3_fetch -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_fetch.log 2>&1
run timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $O/k3_write -o pmc -- python3 tools/k3_sweep.py 6144 > $O/k3_write.log 2>&1
echo done
