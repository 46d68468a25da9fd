#!/bin/bash
# tools/ab_env.sh VAR "v1 v2 ..." [workload] [repeats]  -- A/B of an environment switch on the bench pass, alternating, on one box
VAR=$1; VALS=$2; WL=${3:-era5_1deg_djf30}; REP=${4:-2}
for r in $(seq $REP); do for v in $VALS; do
  env $VAR=$v python bench.py --no-secondary --no-cpu-baseline --no-extra --steps 40 --workload $WL | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$VAR=$v', '$WL', 'pass %.4f thr %.4f rel %.4f' % (d['ms_per_step'], k['k_threshold'], k['k_relabel']))"
done; done
