#!/usr/bin/env bash
# A/B on ONE box: bench.py with the in-tree library and with other builds of it (AZAMD_LIB), alternating, same arguments.
# Box-to-box spread of the headline is +-2-4 %, so two builds are only comparable inside one gpurun call.
# Usage: bash tools/bench_ab.sh "<bench args>" lib1.so lib2.so ...    ("-" = the in-tree library)
set -uo pipefail
ARGS=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset AZAMD_LIB; else export AZAMD_LIB=$lib; fi
    timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline --no-second-leg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-40s games/s %.1f est %.1f ms/step %.1f frac %.3f launch_us %.1f cohorts %d' % ('$lib', d['value'], d['value_steady_state_estimate'], d['ms_per_step'], r['frac'], 1e3 * r['avg_launch_ms'], r['concurrent_cohorts']))
" || exit 1
  done
done
