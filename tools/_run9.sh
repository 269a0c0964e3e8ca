python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "roi" 2>&1 | tail -3
for c in resnet50_voc resnet50_coco2017 vgg16_voc; do
python tools/bench_roi.py $c 2>/dev/null | tail -1 | cut -c1-300
CIM_ROI_FWD_NOSLICE=1 python tools/bench_roi.py $c 2>/dev/null | tail -1 | cut -c1-300
done
python tools/bench_roi_bwd.py 2>/dev/null | cut -c1-400
