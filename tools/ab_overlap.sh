for v in 1 0 1 0 1 0; do
  echo -n "overlap $v: "
  NNHIP_PREPARE_OVERLAP=$v python bench.py --no-cpu-baseline --steps 40 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
