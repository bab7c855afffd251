"""rpvg_amd — MI355X-native engine for rpvg's EM abundance / haplotype-posterior hot path.

The product is native: HIP kernels + C ABI in ``rpvg_amd/csrc`` (librpvg_hip.so)
and the C++ host classes that keep rpvg's ``PathEstimator`` interface in
``rpvg_amd/host`` (librpvg_amd_host.so).  The Python modules here are harness
plumbing only (ctypes bindings, flat batches, process-per-GPU sharding).
"""

__version__ = "0.1.0"
