#!/usr/bin/env bash
# Games per wave of the fused / pure tree searches against the number of games: the same games on
# 1x / 2x / 4x as many waves (AZ_TREE_GPW = full / half / quarter of 64 / LG), from the latency-bound regime (4 096 games) into
# the saturated one (16 384 .. 65 536 games).  One line per (workload, games, GPW): launch time and playouts/s.
# Usage (GPU box, repo root): bash tools/gpw_sweep.sh > gpurun_out/r3_gpw.log
set -uo pipefail
row() {  # label, gpw, bench_configs args...
  local label=$1 gpw=$2; shift 2
  AZ_TREE_GPW=$gpw timeout -k 10 240 python3 tools/bench_configs.py "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('%-28s gpw %d  launch %8.3f ms  %7.1f M playouts/s  depth %.2f' % (d['bench'], $gpw, d['avg_launch_ms'], d['playouts_per_s'] / 1e6, d['mean_depth']))
" || exit 1
}
for g in 4096 16384 65536; do for w in 4 2 1; do row ttt $w ttt --ttt-games $g --mean-len 7.5 || exit 1; done; done
for g in 4096 8192 16384; do for w in 4 2 1; do row nbttt $w nbttt --nbttt-games $g --mean-len 41 || exit 1; done; done
for g in 4096 16384; do for w in 8 4 2; do row pure $w pure --pure-games $g --no-cpu || exit 1; done; done
echo done
