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
    """Eight hardware queues instead of the HIP runtime's four (the library asks for the same when it is loaded; measured for two host
    lanes over one batch and for the batch pipeline's four single-lane engines: rpvg_amd/csrc/context.hip).  The runtime reads the
    variable when it starts, so this has to run before the process touches the GPU: import rpvg_amd first.  A value already in
    the environment is kept."""
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


_ask_for_hardware_queues()


def _confine_to_cpus():
    """RPVG_AMD_CPUS=n: the process keeps to n CPUs, a contiguous block around the one it is on (threads started later inherit
    it).  A knob, not a default: a cgroup CPU quota far below the host's hardware threads (16 CPUs' worth of time on a 256-thread
    box) is handed out in slices per CPU a thread runs on, and a hundred threads wandering over 256 CPUs can be throttled — whole
    periods of tens of milliseconds, the 15 - 20 ms steps among 9.7 ms ones — while they use two thirds of the quota.  On one
    box 32 CPUs took the throttled periods of three 20-step runs from 11 to 2; on the next the unconfined runs had none and were
    the faster ones (means 9.6 - 10.0 against 9.9 - 10.1 ms per batch)."""
    import ctypes
    import os
    try:
        n = int(os.environ.get("RPVG_AMD_CPUS", "0"))
        allowed = sorted(os.sched_getaffinity(0))
        if n <= 0 or n >= len(allowed):
            return
        here = ctypes.CDLL("libc.so.6").sched_getcpu()
        at = allowed.index(here) if here in allowed else 0
        first = max(0, min(len(allowed) - n, (at // n) * n))
        os.sched_setaffinity(0, allowed[first:first + n])
    except (OSError, ValueError):
        pass


_confine_to_cpus()
