cd $GRAFT_REPO_ROOT
for v in 6 7 5 6 7; do
  echo "== VAMD_NOISE_TEAMS=$v"
  VAMD_NOISE_TEAMS=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-neighbours --no-parity-sample 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']/1e6,3), round(d['ms_per_step'],3), {k:round(v,3) for k,v in r['kernels_ms_per_step'].items()})"
done
