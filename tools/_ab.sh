# same-box A/B: another checkout (_ab_base) vs this tree, interleaved runs on ONE gpurun box.
#   git worktree add _ab_base <rev> && (cd _ab_base && python -m cim_amd.build)   # e.g. <rev> = the previous round's last commit
#   gpurun -- 'bash tools/_ab.sh 4'      (remove the worktree afterwards: git worktree remove --force _ab_base)
N=${1:-3}
for i in $(seq $N); do
(cd _ab_base && python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', round(d['ms_per_step'],3))")
python bench.py --no-cpu-baseline --no-extra --phases 16 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('new ', round(d['ms_per_step'],3), {k[:14]: round(v,2) for k,v in d['extra']['phases'].items()})"
done
