mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_gemm_pair.py --no-old --json $R/gpurun_out/r4/gemm_pair_ablation.json > $R/gpurun_out/r4/gemm_pair_ablation.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -o r -- python3 $R/tools/bench_gemm_pair.py --no-old --no-alts --only "pair wino,pair fc1" > $R/gpurun_out/r4/pmc_mfma.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmc_mfma gemm_pair_kernel > $R/gpurun_out/r4/pmc_mfma_gemm_pair.json
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_roi -o r -- python3 $R/tools/bench_roi.py > $R/gpurun_out/r4/pmc_roi.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmc_roi roi_align roi_partial roi_tables > $R/gpurun_out/r4/pmc_roi_align.json
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_roib -o r -- python3 $R/tools/bench_roi_bwd.py > $R/gpurun_out/r4/pmc_roib.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmc_roib roi_align_bwd roi_partial > $R/gpurun_out/r4/pmc_roi_align_bwd.json
cat $R/gpurun_out/r4/gemm_pair_ablation.txt | tail -22
python3 -c "
import json
for f in ('pmc_mfma_gemm_pair','pmc_roi_align','pmc_roi_align_bwd'):
    d=json.load(open('$R/gpurun_out/r4/%s.json'%f))
    for k,v in d.items():
        print(k[:70], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!='counters'})
"
