#!/usr/bin/env bash
# bench.py over a matrix of tuning switches on ONE box (box-to-box spread is larger than most effects).
# Usage: bash tools/bench_matrix.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...     (one run per env string; "-" = none)
set -uo pipefail
ARGS=$1; shift
for envs in "$@"; do
  [ "$envs" = "-" ] && envs=""
  env $envs timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline --no-second-leg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; c = d['config']
print('%-44s games/s %7.1f est %7.1f ms/step %6.1f Mplayouts/s %5.1f evals/playout %.3f leaves/launch %4.0f K3 launch_us %5.1f busy %.2f frac %.3f tree_us %5.1f playouts/launch %4.0f' % ('$envs' + ' ' + ' '.join(a for a in '''$ARGS'''.split() if a.startswith('--co') or a.isdigit())[:0], d['value'], d['value_steady_state_estimate'], d['ms_per_step'], d['sims_per_sec'] / 1e6, c['leaf_evals'] / max(d['sims_per_sec'] * d['ms_per_step'] * d['steps'] / 1e3, 1), r.get('leaves_per_launch', 0), 1e3 * r['avg_launch_ms'], r.get('busy_ms', 0) / (d['ms_per_step'] * d['steps']) / max(r.get('timed_launches', 1) / max(r.get('launches', 1), 1), 1e-9), r['frac'], 1e3 * (d.get('roofline_tree_step') or {}).get('avg_launch_ms', 0), (d.get('roofline_tree_step') or {}).get('playouts_per_launch', 0)))
" || exit 1
done
