cd $GRAFT_REPO_ROOT
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extra --phases 16 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k[:14]: round(v,2) for k,v in d['extra']['phases'].items()}); print([(list(r.values())[0][:40], r.get('ms')) for r in d['roofline_hbm']])"
done
