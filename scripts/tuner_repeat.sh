# how often does the schedule tuner keep which plan?  N runs of the driver's command, one line each
n=${1:-10}
for rep in $(seq $n); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --hold-seconds 0 --modes off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-8s deal %-14s %8.1f Msamples/s  %s' % (d['config']['scene'], d['deal'], d['value'], d['schedule']))"
done
