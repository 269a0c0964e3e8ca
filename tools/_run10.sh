cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
for i in 1 2 3; do
for lib in "" cim_amd/libcim_hip_alt_wg256.so cim_amd/libcim_hip_alt_wg384.so cim_amd/libcim_hip_alt_wg768s8.so; do
CIM_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extra --phases 16 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-product}'.split('alt_')[-1], round(d['ms_per_step'],3), {k[:14]: round(v,2) for k,v in d['extra']['phases'].items()})"
done; done > gpurun_out/r5b/ab_small.log 2>&1; cat gpurun_out/r5b/ab_small.log
