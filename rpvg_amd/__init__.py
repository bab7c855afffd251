"""rpvg_amd — MI355X-native engine for rpvg's EM abundance / haplotype-posterior hot path.

The product is native: HIP kernels + C ABI in ``rpvg_amd/csrc`` (librpvg_hip.so)
and the C++ host classes that keep rpvg's ``PathEstimator`` interface in
``rpvg_amd/host`` (librpvg_amd_host.so).  The Python modules here are harness
plumbing only (ctypes bindings, flat batches, process-per-GPU sharding).
"""

__version__ = "0.1.0"


def _keep_heap_top():
    """glibc hands the top of every heap back to the kernel when a batch's temporaries are dropped and faults it in
    again for the next batch (45 of 58 CPU-seconds of 60 bench batches were system time).  A 64 MB top pad keeps it;
    it has to be set before the host threads (and their malloc arenas) exist, so at import.  MALLOC_TOP_PAD_ wins.
    The C++ host does the same in HipEngine (rpvg_amd/host/hip_engine.cpp)."""
    import ctypes
    import os
    if "MALLOC_TOP_PAD_" in os.environ:
        return
    try:
        ctypes.CDLL("libc.so.6").mallopt(-2, 64 << 20)  # M_TOP_PAD
    except OSError:
        pass


_keep_heap_top()


def _ask_for_hardware_queues():
    """Eight hardware queues instead of the HIP runtime's four (two host lanes x four busy streams; see
    rpvg_amd/csrc/common.hpp).  The runtime reads the variable when it starts, so this has to run before the process
    touches the GPU: import rpvg_amd first.  A value already in the environment is kept."""
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


_ask_for_hardware_queues()
