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
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

namespace rpvg_amd {

class PipelineWorker {

    public:

        PipelineWorker() : stopping(false), has_task(false), busy(false), worker(&PipelineWorker::loop, this) {}

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

        std::thread worker;
};

}

#endif
