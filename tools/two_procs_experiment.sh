export AZ_ROLL_CAP=6 AZ_TREE_STEP_GPW=2
for n in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cohorts 4 --games 1024 --no-cpu-baseline --no-second-leg --no-profile > gpurun_out/r3_two_$n.json 2>/dev/null &
done
wait
for n in 1 2; do python3 -c "
import sys, json
d = json.loads(open('gpurun_out/r3_two_$n.json').read().strip().splitlines()[-1])
print('proc $n', round(d['value'],1), round(d['ms_per_step'],1), round(d['sims_per_sec']/1e6,1))
"; done
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cohorts 4 --games 1024 --no-cpu-baseline --no-second-leg --no-profile 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('single 1024', round(d['value'],1), round(d['ms_per_step'],1), round(d['sims_per_sec']/1e6,1))
"
