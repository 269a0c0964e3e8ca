#!/bin/bash
# Regenerates the inputs of profiles/rN on the GPU box (run through gpurun; writes under gpurun_out/refresh/).
#   bench line (with cpu_baseline and --phases), rocprofv3 kernel trace + stats of the same command (one row per GEMM PRODUCT),
#   two --pmc passes for HBM traffic (separate runs) + their calibration passes, an SQ / GRBM pass for MFMA-busy and the
#   wave-cycle split of the dominant GEMM and of the ROIAlign kernels, the GEMM ablation table, the other configurations,
#   the micro-benchmarks.   Ablation builds: tools/build_alt.sh exp3|exp4|exp5 gemm_pair.hip -DCIM_PAIR_EXP=3|4|5 beforehand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --phases 24 > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o r -- python3 "$ROOT/bench.py" --steps 8 --warmup 8 --no-cpu-baseline --no-extra > "$OUT/trace_bench.log" 2>&1
python3 "$ROOT/tools/trace_steps.py" /tmp/kt/r_kernel_trace.csv 8 > "$OUT/bench_kernel_stats.csv"
python3 "$ROOT/tools/trace_phase.py" /tmp/kt/r_kernel_trace.csv > "$OUT/backbone_phases.txt"
python3 "$ROOT/tools/trace_tail.py" /tmp/kt/r_kernel_trace.csv > "$OUT/step_tail.txt"
python3 "$ROOT/tools/trace_phases_step.py" /tmp/kt/r_kernel_trace.csv > "$OUT/step_phases_profiled.txt"
python3 "$ROOT/tools/trace_timeline.py" /tmp/kt/r_kernel_trace.csv --min-us 15 > "$OUT/step_timeline.txt"
cp /tmp/kt/r_kernel_stats.csv "$OUT/rocprofv3_kernel_stats_uncut.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o r -- python3 "$ROOT/bench.py" --steps 8 --warmup 8 --no-cpu-baseline --no-extra > "$OUT/pmc_$c.log" 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -o r -- python3 "$ROOT/tools/pmc_calibrate.py" > "$OUT/cal_$c.log" 2>&1
done
python3 "$ROOT/tools/pmc_traffic.py" /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE winograd7 f16x2p mix8 --calibrate /tmp/cal_FETCH_SIZE /tmp/cal_WRITE_SIZE > "$OUT/pmc_traffic_resnet50_voc.json"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/pmc_mfma -o r -- python3 "$ROOT/tools/bench_gemm_pair.py" --no-old --no-alts --only "pair wino,pair fc1,form1" > "$OUT/pmc_mfma.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmc_mfma gemm_pair_kernel > "$OUT/pmc_mfma_gemm_pair.json"
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmc_mfma gemm_pair_ring_kernel > "$OUT/pmc_mfma_gemm_pair_ring.json"
SQR="SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQR --output-format csv -d /tmp/pmc_roi -o r -- python3 "$ROOT/tools/bench_roi.py" > "$OUT/pmc_roi.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmc_roi roi_align roi_partial roi_tables wino7 > "$OUT/pmc_sq_roi_align.json"
rocprofv3 --kernel-trace --pmc $SQR --output-format csv -d /tmp/pmc_roib -o r -- python3 "$ROOT/tools/bench_roi_bwd.py" > "$OUT/pmc_roib.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmc_roib roi_align_bwd roi_partial > "$OUT/pmc_sq_roi_align_bwd.json"
python3 "$ROOT/tools/bench_gemm_pair.py" --no-old --json "$OUT/gemm_pair_ablation.json" > "$OUT/gemm_pair_ablation.txt" 2>&1
for cfg in vgg16_voc resnet50_coco2017 hrnet48_coco2017; do
  python3 "$ROOT/bench.py" --config $cfg --steps 8 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | tail -1
done > "$OUT/bench_other_configs.json"
python3 "$ROOT/bench.py" --fixed-image --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > "$OUT/bench_fixed_image.json"
python3 "$ROOT/tools/bench_roi_bwd.py" 2>/dev/null | grep "^{" > "$OUT/bench_roi_bwd.json"
python3 "$ROOT/tools/bench_roi.py" 2>/dev/null | tail -1 > "$OUT/bench_roi.json"
python3 "$ROOT/tools/bench_mining.py" > "$OUT/bench_mining.txt" 2>&1
python3 "$ROOT/tools/bench_mining.py" --config resnet50_coco2017 >> "$OUT/bench_mining.txt" 2>&1
python3 "$ROOT/tools/bench_conv3x3.py" 2>/dev/null | grep "^{" > "$OUT/bench_conv3x3.json"
python3 "$ROOT/tools/bench_gemm_small.py" 2>/dev/null | grep "^{" > "$OUT/bench_gemm_small.json"
# round 6: SQ counters of the backbone's kernels (the micro-benchmarks above under --pmc), per-dispatch-run GPU durations of the same
SQB="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d /tmp/pmc_gs -o r -- python3 "$ROOT/tools/bench_gemm_small.py" > "$OUT/pmc_gs.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmc_gs gemm_small small_splitk > "$OUT/pmc_sq_backbone_gemm_small.json"
rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d /tmp/pmc_c3 -o r -- python3 "$ROOT/tools/bench_conv3x3.py" > "$OUT/pmc_c3.log" 2>&1
python3 "$ROOT/tools/pmc_summary.py" /tmp/pmc_c3 conv3x3_small small_splitk > "$OUT/pmc_sq_backbone_conv3x3.json"
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_gs -o r -- python3 "$ROOT/tools/bench_gemm_small.py" > /dev/null 2>&1
python3 "$ROOT/tools/trace_by_dispatch.py" /tmp/kt_gs/r_kernel_trace.csv 20 > "$OUT/gemm_small_kernel_trace.txt"
# same-box A/B of the round's scheduling changes (interleaved)
for rep in 1 2 3; do
  for flag in "" "--no-overlap-update" "--dw-form 0"; do
    python3 "$ROOT/bench.py" $flag --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.load(sys.stdin); print('${flag:-as shipped}', round(d['ms_per_step'],3))"
  done
done > "$OUT/schedule_ab_refresh.txt"
for rep in 1 2 3; do
  for flag in "--dw-form 1" "--dw-form 0"; do python3 "$ROOT/tools/diag_late.py" $flag 2>/dev/null | tail -8; done
done > "$OUT/last_backward_phase_by_stream.txt"
(cd "$ROOT" && python3 -m pytest tests/test_gpu_tolerance.py tests/test_gpu_fullsize.py -q -m gpu > "$OUT/parity_tests.log" 2>&1; cp gpurun_out/parity_deviation.json "$OUT/parity_deviation.json" 2>/dev/null)
ls -la "$OUT"
