#!/bin/bash
# A / B of an environment switch inside ONE gpurun call: tools/ab_env.sh VAR "bench arguments" [rounds]  -> ms_per_step per setting
var=$1; args=$2; n=${3:-2}
for i in $(seq $n); do
  for v in 0 1; do
    env $var=$v python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$var=$v', round(d['ms_per_step'],4), d.get('stages_us'))"
  done
done
