python -m pytest tests/test_gpu_gemm.py tests/test_gpu_dp.py tests/test_e2e_reference.py "tests/test_gpu_parity.py::test_training_step_matches_cpu_oracle" -x -q -m gpu 2>&1 | tail -3
bash tools/_ab.sh 3
