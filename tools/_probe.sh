cd $GRAFT_REPO_ROOT
for rep in 1 2; do
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-neighbours 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernels_ms_per_step'], d.get('parity_sample',{}).get('mismatches'))"
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/p; timeout 300 rocprofv3 --pmc $ctr -d /tmp/p -o x -- python $GRAFT_REPO_ROOT/tools/prof_run.py 131072 1 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py pmc /tmp/p/x_results.db 2>&1 | grep -i "tone\|k_floor\|calib"
done
