// Two host lanes over one GPU.
//
// A batch step alternates host phases (grouping haplotypes, selecting path subsets, merging results)
// with device phases (the calls into the C ABI block until the GPU is done).  Clusters are independent
// units of inference (the reference hands them to OpenMP threads one by one, src/main.cpp:829), so a
// batch can be cut in two and the halves run on two host threads: while one lane waits for the GPU
// the other does its host work; the engine serialises the device calls themselves.  The second lane
// is a persistent worker (a new thread per batch would rebuild its OpenMP team every time).
#ifndef RPVG_AMD_PIPELINE_LANES_HPP
#define RPVG_AMD_PIPELINE_LANES_HPP

#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "experiments.hpp"

namespace rpvg_amd {

// Staggered start of the lanes of one batch.  Lanes that begin together stay in lock step — the same host
// phase at the same time, then the GPU at the same time — and nothing overlaps.  Lane i therefore begins
// only when lane i - 1 has handed its first device stage to the GPU (passBaton), which shifts the lanes by
// one host phase against each other: from then on one lane's host work runs under the other's kernels.
class LaneStagger {

    public:

        explicit LaneStagger(const int num_lanes) : released(num_lanes, false) {

            released.at(0) = true;

            // RPVG_AMD_NO_STAGGER=1 (A/B knob): every lane starts at once
            static const bool no_stagger = RPVG_AMD_EXPERIMENT_ENV("RPVG_AMD_NO_STAGGER") != nullptr;

            if (no_stagger) {

                released.assign(num_lanes, true);
            }
        }

        // Blocks lane until the lane before it has passed the baton.
        void waitTurn(const int lane) {

            std::unique_lock<std::mutex> lock(mutex);
            changed.wait(lock, [&] { return released.at(lane); });
        }

        // Lets the next lane start (idempotent; also called when a lane ends, so that a lane that fails or has
        // nothing to hand over cannot block the others).
        void passBaton(const int lane) {

            {
                std::lock_guard<std::mutex> lock(mutex);

                if (static_cast<size_t>(lane) + 1 < released.size()) {

                    released.at(lane + 1) = true;
                }
            }

            changed.notify_all();
        }

    private:

        std::mutex mutex;
        std::condition_variable changed;
        std::vector<bool> released;
};

class PipelineWorker {

    public:

        // idle_work_in: run by the worker thread after every task, once the task has been reported done (the lanes drop
        // the containers they retired there: nobody waits for it, and nothing is left for the end of the process)
        explicit PipelineWorker(std::function<void()> idle_work_in = nullptr) : stopping(false), has_task(false), busy(false), idle_work(std::move(idle_work_in)), worker(&PipelineWorker::loop, this) {}

        ~PipelineWorker() {

            {
                std::lock_guard<std::mutex> lock(mutex);
                stopping = true;
            }

            wake.notify_all();
            worker.join();
        }

        PipelineWorker(const PipelineWorker &) = delete;
        PipelineWorker & operator=(const PipelineWorker &) = delete;

        // Starts task on the worker thread; one task at a time.
        void submit(std::function<void()> task_in) {

            std::unique_lock<std::mutex> lock(mutex);
            done.wait(lock, [this] { return !busy && !has_task; });

            task = std::move(task_in);
            error = nullptr;
            has_task = true;

            lock.unlock();
            wake.notify_all();
        }

        // Waits for the submitted task; rethrows what it threw.
        void wait() {

            std::unique_lock<std::mutex> lock(mutex);
            done.wait(lock, [this] { return !busy && !has_task; });

            if (error) {

                std::exception_ptr thrown = error;
                error = nullptr;
                std::rethrow_exception(thrown);
            }
        }

    private:

        void loop() {

            std::unique_lock<std::mutex> lock(mutex);

            while (true) {

                wake.wait(lock, [this] { return stopping || has_task; });

                if (stopping) {

                    return;
                }

                std::function<void()> current = std::move(task);
                has_task = false;
                busy = true;
                lock.unlock();

                std::exception_ptr thrown = nullptr;

                try {

                    current();

                } catch (...) {

                    thrown = std::current_exception();
                }

                lock.lock();
                error = thrown;
                busy = false;
                done.notify_all();

                if (idle_work) {

                    lock.unlock();

                    try {

                        idle_work();

                    } catch (...) {
                    }

                    lock.lock();
                }
            }
        }

        std::mutex mutex;
        std::condition_variable wake;
        std::condition_variable done;

        bool stopping;
        bool has_task;
        bool busy;

        std::function<void()> task;
        std::exception_ptr error;
        std::function<void()> idle_work;

        std::thread worker;
};

// Marks the calling thread as inside a lane of PathEstimator::runInLanes: only there is somebody who drops what
// RetiredContainers keeps (the lane itself, once its work is done).  Outside of it, owners free on the spot.
class LaneScope {

    public:

        LaneScope() { ++depth(); }
        ~LaneScope() { --depth(); }

        LaneScope(const LaneScope &) = delete;
        LaneScope & operator=(const LaneScope &) = delete;

        static bool active() { return depth() > 0; }

    private:

        static int & depth() {

            thread_local int scopes = 0;
            return scopes;
        }
};

// Containers a lane is done with, kept until the lane has time to drop them.
//
// A batch leaves tens of thousands of small vectors and maps behind (EM problems and solutions, group posteriors,
// subset weights); freeing them takes the last lane 0.7 ms at the very end of the batch and 0.3 ms between its search
// and its EM, with the GPU idle.  A lane other than the first starts every batch waiting for the lane before it: its
// worker thread drops the previous batch's containers then (dropAll() in PathEstimator::runInLanes).  The first
// lane frees on the spot: it never waits, and it finishes before the others anyway.
class RetiredContainers {

    public:

        // of the calling host thread (= lane)
        static RetiredContainers & ofThisThread() {

            thread_local RetiredContainers retired;
            return retired;
        }

        // `drop` owns the containers (captures them by shared_ptr) and frees their elements when called
        void keep(std::function<void()> drop) {

            pending.emplace_back(std::move(drop));
        }

        void dropAll() {

            for (auto & drop: pending) {

                drop();
            }

            pending.clear();
        }

        ~RetiredContainers() {

            pending.clear();  // the captured containers go with their closures
        }

    private:

        std::vector<std::function<void()> > pending;
};

}

#endif
