run() { python bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['value'],2))"; }
run --iter-size 4 --steps 8
run --iter-size 1
(cd _ab_base && run --iter-size 4 --steps 8)
run --iter-size 4 --steps 8
(cd _ab_base && run --iter-size 4 --steps 8)
