"""Horizon of the `lap8` evolve() comparison on the GPU (tests/test_gpu_evolve.py), justified on the CPU by
tests/test_rounding_sensitivity.py::test_lap8_evolve_horizon_follows_from_the_perturbed_oracle: the oracle perturbed by one
ulp per entry of H keeps bookkeeping and survivor order for >= 45 iterations; 8 ulps cost ~4 iterations of that."""
EVOLVE_LAP8_ITERS = 40
