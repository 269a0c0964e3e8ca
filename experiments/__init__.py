"""Superseded arithmetic engines and convolution algorithms of the MaskFuse contractions (rounds 1-3): bf16x3, f16x2 and true-fp32 MFMA
GEMMs, direct / F(2x2,3x3) / F(4x4,3x3) convolution.  TEST INFRASTRUCTURE: `tests/test_gpu_tolerance.py` measures every engine x
algorithm combination against the reference, `tests/test_gpu_experiments.py` checks the kernels themselves.  Nothing under `cim_amd/`
imports this package; the product has one engine (the pair engine, cim_amd/csrc/gemm_pair.hip) and one algorithm (the mixed 4 + 3
Winograd tiling).  Built by `python -m experiments.build` (and by `__graft_entry__.build()`) into experiments/libcim_exp.so."""
