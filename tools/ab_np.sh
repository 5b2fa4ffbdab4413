# A/B of library builds on one box: tools/ab_np.sh lib1.so lib2.so ...   (python bench.py --steps 30 each, twice)
for r in 1 2; do for v in "$@"; do
echo -n "$v: "
NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=$v python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
k=d.get('kernel_classes',{})
print(d['value'], d['ms_per_step'], 'box', d['box100k']['ms_per_step'], 'train_large', d['train_large']['ms_per_step'], 'train_small', d['train_small']['ms_per_step'], 'strong8', d['strong_projection']['by_n_gpus']['8']['ms_per_step'], {c: round(v['ms_per_step'],3) for c,v in k.items() if c.startswith('edge_')})"
done; done
