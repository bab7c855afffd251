// Optional wall-clock phase tracing of the host layer (RPVG_AMD_TRACE=1):
// the reference only has stage timers around whole pipeline stages
// (src/main.cpp:612-1091); these split the inference stage itself.
#ifndef RPVG_AMD_TRACE_HPP
#define RPVG_AMD_TRACE_HPP

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include <omp.h>
#include <time.h>

#include "experiments.hpp"

namespace rpvg_amd {

// Threads used by the host-side parallel loops (flattening, subset selection, merging).  These loops
// are short; a modest team avoids waking (and then spinning) every hardware thread of a large host
// between GPU calls.  RPVG_AMD_HOST_THREADS overrides.
// Team size override of the calling host thread (0 = none): the two lanes of a pipelined batch
// (pipeline_lanes.hpp) share the host's threads.
inline int & hostThreadsOverride() {

    thread_local int override_threads = 0;
    return override_threads;
}

inline int hostThreads() {

    if (hostThreadsOverride() > 0) {

        return hostThreadsOverride();
    }

    static const int threads = []() {

        if (const char * env = std::getenv("RPVG_AMD_HOST_THREADS")) {

            return std::max(1, std::atoi(env));
        }

        // 32: the parallel loops are short; larger teams bring no time (measured 32 vs 48 over 150 batches: equal) and
        // burn a third more CPU (0.19 vs 0.13-0.15 CPU-seconds per batch: on a host that grants the process 16 CPUs'
        // worth of time that is the quota)
        int threads = std::min(32, omp_get_max_threads());

        // one process per GPU: share the host's hardware threads between the local ranks
        int local_ranks = 1;

        if (const char * local_world = std::getenv("LOCAL_WORLD_SIZE")) {

            local_ranks = std::max(1, std::atoi(local_world));
            threads = std::min(threads, std::max(4, omp_get_max_threads() / local_ranks));
        }

        // ... and the CPU time the cgroup grants (cpu.max: quota and period): a team may burst to twice its share of it
        // (one rank on a 16-CPU quota: 32 threads, the measured optimum; eight ranks on the same quota: 4 each instead of
        // 8 x 2 lanes x 32 threads that would spend the quota in the first millisecond of every period)
        if (FILE * cpu_max = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {

            long long quota = 0, period = 0;

            if (std::fscanf(cpu_max, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {

                const int share = static_cast<int>((2 * quota) / (period * local_ranks));
                threads = std::min(threads, std::max(4, share));
            }

            std::fclose(cpu_max);
        }

        return std::max(1, threads);
    }();

    return threads;
}

// The team of a loop that is over in a fraction of a millisecond whatever its size (filling a few thousand small containers, 5 000
// generator states): waking a whole team for it costs more CPU time than the loop.  A quarter of the team, at least two threads:
// configs[4] through the pipeline (four estimator threads, eight host threads each), five such loops per batch — 41-46 ms of CPU per
// batch with the whole teams, 33-36 with two threads each, 29 with one, the step unchanged (12.1-13.4 / 12.6-12.7 / 13.1 ms).
inline int shortLoopThreads() {

    return std::min(hostThreads(), std::max(2, hostThreads() / 4));
}

class PhaseTrace {

    public:

        static bool enabled() {

            static const bool on = (std::getenv("RPVG_AMD_TRACE") != nullptr);
            return on;
        }

        static void add(const std::string & phase, const double seconds) {

            std::lock_guard<std::mutex> lock(mutex());
            totals()[phase] += seconds;
        }

        // RPVG_AMD_TRACE_CPU (with RPVG_AMD_TRACE): next to a phase's wall time the CPU time the WHOLE PROCESS spent while it was
        // open ("<phase> [process CPU]": the phase's own threads, its OpenMP team — spinning included — and whoever else ran;
        // meaningful with one call in flight)
        static bool cpu() {

            static const bool on = enabled() && (std::getenv("RPVG_AMD_TRACE_CPU") != nullptr);
            return on;
        }

        static double processCpuSeconds() {

            timespec ts;
            clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
            return static_cast<double>(ts.tv_sec) + 1e-9 * static_cast<double>(ts.tv_nsec);
        }

        // RPVG_AMD_TIMELINE: every phase with its start and end (ms since the first phase of the process)
        // and the OpenMP-independent id of the host thread that ran it — who waited for whom.
        static bool timeline() {

            static const bool on = (std::getenv("RPVG_AMD_TIMELINE") != nullptr);
            return on;
        }

        static void event(const char * phase, const std::chrono::steady_clock::time_point start, const std::chrono::steady_clock::time_point end) {

            static const std::chrono::steady_clock::time_point origin = start;
            static std::atomic<int> next_thread(0);
            thread_local const int thread_id = next_thread++;

            std::fprintf(stderr, "[timeline] thread %d %-44s %9.3f %9.3f\n", thread_id, phase, std::chrono::duration<double, std::milli>(start - origin).count(), std::chrono::duration<double, std::milli>(end - origin).count());
        }

        static void report() {

            if (!enabled()) {

                return;
            }

            std::lock_guard<std::mutex> lock(mutex());

            for (auto & phase: totals()) {

                std::fprintf(stderr, "[rpvg_amd trace] %-40s %9.3f ms\n", phase.first.c_str(), phase.second * 1e3);
            }

            totals().clear();
        }

    private:

        static std::mutex & mutex() { static std::mutex m; return m; }
        static std::map<std::string, double> & totals() { static std::map<std::string, double> t; return t; }
};

class ScopedPhase {

    public:

        explicit ScopedPhase(const char * name_in) : name(name_in), start(std::chrono::steady_clock::now()), cpu_start(PhaseTrace::cpu() ? PhaseTrace::processCpuSeconds() : 0.0) {}

        ~ScopedPhase() {

            stop();
        }

        // Ends the phase before the end of the scope.
        void stop() {

            if (!stopped && PhaseTrace::enabled()) {

                PhaseTrace::add(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count());

                if (PhaseTrace::cpu()) {

                    PhaseTrace::add(std::string(name) + " [process CPU]", PhaseTrace::processCpuSeconds() - cpu_start);
                }
            }

            if (!stopped && PhaseTrace::timeline()) {

                PhaseTrace::event(name, start, std::chrono::steady_clock::now());
            }

            stopped = true;
        }

    private:

        const char * name;
        const std::chrono::steady_clock::time_point start;
        const double cpu_start;
        bool stopped = false;
};

}

#endif
