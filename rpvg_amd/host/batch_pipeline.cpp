#include "batch_pipeline.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>

#include "estimator_factory.hpp"
#include "trace.hpp"

namespace rpvg_amd {

// Four batches side by side; six where the posteriors are sampled (--use-hap-gibbs): the device sampler is driven in chunks of rounds
// with a look at its progress between them, and a batch spends more of its time waiting for that than the EM models' batches do
// (configs[4]: 10.1 against 11.6-11.8 ms per batch, 16.4 with eight; configs[2]: 4.3-4.5 with four, five or six).
static int defaultWorkers(const rpvg_params & params) {

    const char * env = std::getenv("RPVG_AMD_PIPELINE_WORKERS");
    return env ? std::max(1, std::min(8, std::atoi(env))) : (params.use_hap_gibbs ? 6 : 4);
}

BatchPipeline::BatchPipeline(const int device_in, const std::string & model_in, const rpvg_params & params_in, const int workers) : device(device_in), model(model_in), params(params_in), num_resident(0), num_unfinished(0), stopping(false), first_error(nullptr), upload_seconds(0), finish_seconds(0), estimate_seconds(0), wait_for_batch_seconds(0), upload_batches(0), stats_epoch(std::chrono::steady_clock::now()) {

    const int num_workers = workers > 0 ? std::min(workers, 8) : defaultWorkers(params_in);

    // (engines first, on the calling thread: a failure — no GPU — is the constructor's)
    // (two uploaders: the copies of one batch run while the other's kernels, wait and bookkeeping do — the link stays busy)
    static const int num_uploaders = []() {

        const char * env = std::getenv("RPVG_AMD_PIPELINE_UPLOADERS");
        return env ? std::max(1, std::min(4, std::atoi(env))) : 1;
    }();

    for (int uploader = 0; uploader < num_uploaders; ++uploader) {

        uploader_engines.emplace_back(std::make_shared<HipEngine>(device, true));
    }

    // batches on the GPU at any time: one per worker, one per uploader, and RPVG_AMD_PIPELINE_SPARE waiting for a worker (default 1)
    static const int spare = []() {

        const char * env = std::getenv("RPVG_AMD_PIPELINE_SPARE");
        return env ? std::max(0, std::min(8, std::atoi(env))) : 1;
    }();

    max_resident = num_workers + num_uploaders + spare;

    for (int worker = 0; worker < num_workers; ++worker) {

        worker_engines.emplace_back(std::make_shared<HipEngine>(device, false, 1));
    }

    makePathEstimator(model, params, worker_engines.front());  // (an unknown model name throws here, not on a worker)

    for (int uploader = 0; uploader < num_uploaders; ++uploader) {

        upload_threads.emplace_back(&BatchPipeline::uploadLoop, this, uploader);
    }

    // (RPVG_AMD_FINISH_ON_ESTIMATOR=1: the estimators run the kernels behind the copies themselves, as in the first half of round 5)
    static const bool finish_on_estimator = std::getenv("RPVG_AMD_FINISH_ON_ESTIMATOR") != nullptr && std::atoi(std::getenv("RPVG_AMD_FINISH_ON_ESTIMATOR")) != 0;

    if (!finish_on_estimator && num_uploaders == 1) {

        queue_thread = std::thread(&BatchPipeline::queueLoop, this);
        ++max_resident;
    }

    for (int worker = 0; worker < num_workers; ++worker) {

        worker_threads.emplace_back(&BatchPipeline::workerLoop, this, worker);
    }
}

BatchPipeline::~BatchPipeline() {

    {
        std::lock_guard<std::mutex> lock(mutex);
        stopping = true;
    }

    changed.notify_all();

    for (auto & thread: upload_threads) {

        thread.join();
    }

    if (queue_thread.joinable()) {

        queue_thread.join();
    }

    for (auto & thread: worker_threads) {

        thread.join();
    }

    // what is left (a pipeline torn down with batches in it) goes with the uploader's engine still alive
    to_upload.clear();
    copied.clear();
    resident.clear();
}

void BatchPipeline::submit(const rpvg_cluster_batch & host_batch, std::vector<PathClusterEstimates> * estimates) {

    if (!estimates || estimates->size() != host_batch.num_clusters) {

        throw EngineError("BatchPipeline::submit: one PathClusterEstimates per cluster of the batch is required");
    }

    std::unique_ptr<Job> job(new Job());
    job->host_batch = host_batch;
    job->estimates = estimates;

    std::unique_lock<std::mutex> lock(mutex);

    // the containers are written by the batch that has them: a second one on the same containers takes its turn
    changed.wait(lock, [&] { return first_error || busy_estimates.count(estimates) == 0; });

    if (first_error) {

        return;  // (wait() reports it)
    }

    busy_estimates.insert(estimates);
    ++num_unfinished;
    to_upload.emplace_back(std::move(job));

    lock.unlock();
    changed.notify_all();
}

void BatchPipeline::wait() {

    std::unique_lock<std::mutex> lock(mutex);
    changed.wait(lock, [&] { return num_unfinished == 0; });

    if (first_error) {

        std::exception_ptr error = first_error;
        first_error = nullptr;
        std::rethrow_exception(error);
    }
}

std::vector<std::unique_ptr<BatchPipeline::Job> > BatchPipeline::fail(std::exception_ptr error) {

    // (mutex held) the batches that have not started are dropped: their containers stay as they are.  Only the bookkeeping
    // happens here: freeing a device batch takes its context's lock and waits for its copies, and every thread of the pipeline
    // would stand behind this mutex meanwhile.
    if (!first_error) {

        first_error = error;
    }

    std::vector<std::unique_ptr<Job> > dropped;

    for (auto & job: to_upload) {

        busy_estimates.erase(job->estimates);
        --num_unfinished;
        dropped.emplace_back(std::move(job));
    }

    to_upload.clear();

    for (auto & job: copied) {

        busy_estimates.erase(job->estimates);
        --num_unfinished;
        --num_resident;
        dropped.emplace_back(std::move(job));
    }

    copied.clear();

    for (auto & job: resident) {

        busy_estimates.erase(job->estimates);
        --num_unfinished;
        --num_resident;
        dropped.emplace_back(std::move(job));
    }

    resident.clear();

    return dropped;
}

void BatchPipeline::uploadLoop(const int uploader) {

    const auto & uploader_engine = uploader_engines.at(uploader);

    // (the uploads use no team of their own any more; what is left of the host's parallel loops is the workers')
    hostThreadsOverride() = 1;

    while (true) {

        std::unique_ptr<Job> job;

        {
            std::unique_lock<std::mutex> lock(mutex);
            changed.wait(lock, [&] { return stopping || (!to_upload.empty() && num_resident < max_resident); });

            if (stopping) {

                return;
            }

            job = std::move(to_upload.front());
            to_upload.pop_front();
            ++num_resident;
        }

        std::exception_ptr error = nullptr;
        const auto start = std::chrono::steady_clock::now();

        try {

            // (the copies only: the kernels behind them are the worker's — the uploader's next copy starts as soon as this one ends)
            job->device_batch.reset(new DeviceClusterBatch(uploader_engine, job->host_batch, true));


        } catch (...) {

            error = std::current_exception();
        }

        const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();

        std::vector<std::unique_ptr<Job> > dropped;  // (destroyed behind the block: without the mutex)

        {
            std::lock_guard<std::mutex> lock(mutex);

            if (error) {

                busy_estimates.erase(job->estimates);
                --num_unfinished;
                --num_resident;
                dropped = fail(error);

            } else if (first_error) {  // (a batch failed meanwhile: this one is dropped like those behind it)

                busy_estimates.erase(job->estimates);
                --num_unfinished;
                --num_resident;

            } else {

                upload_seconds += seconds;
                ++upload_batches;
                (queue_thread.joinable() ? copied : resident).emplace_back(std::move(job));
            }
        }

        changed.notify_all();
    }
}

// The kernels behind a batch's copies (rpvg_hip_batch_upload_finish_queue) are queued by a thread of their own on the uploader's
// context — its side stream: no context, no hardware queue more — while the uploader queues and waits for the next batch's copies;
// the estimator that takes the batch waits for them (_finish_wait) instead of running them.  Queuing them costs a thread 0.5 ms per
// batch, which the uploader does not have (4.7 against 3.6 ms per upload with both on one thread).
void BatchPipeline::queueLoop() {

    hostThreadsOverride() = 1;

    while (true) {

        std::unique_ptr<Job> job;

        {
            std::unique_lock<std::mutex> lock(mutex);
            changed.wait(lock, [&] { return stopping || !copied.empty(); });

            if (stopping) {

                return;
            }

            job = std::move(copied.front());
            copied.pop_front();
        }

        std::exception_ptr error = nullptr;

        try {

            job->device_batch->queueFinish();

        } catch (...) {

            error = std::current_exception();
        }

        std::vector<std::unique_ptr<Job> > dropped;  // (destroyed behind the block: without the mutex)

        {
            std::lock_guard<std::mutex> lock(mutex);

            if (error || first_error) {  // (this batch failed, or another did meanwhile: dropped like those behind it)

                busy_estimates.erase(job->estimates);
                --num_unfinished;
                --num_resident;

                if (error) {

                    dropped = fail(error);
                }

            } else {

                resident.emplace_back(std::move(job));
            }
        }

        changed.notify_all();
    }
}

void BatchPipeline::workerLoop(const int worker) {

    const auto & engine = worker_engines.at(worker);

    // the host's parallel loops (the models other than the default haplotype-transcripts path still have some) share the
    // rank's threads between the workers
    hostThreadsOverride() = std::max(1, hostThreads() / static_cast<int>(worker_engines.size()));

    std::unique_ptr<PathEstimator> estimator;

    while (true) {

        std::unique_ptr<Job> job;
        const auto idle_from = std::chrono::steady_clock::now();

        {
            std::unique_lock<std::mutex> lock(mutex);
            changed.wait(lock, [&] { return stopping || !resident.empty(); });

            if (stopping) {

                return;
            }

            job = std::move(resident.front());
            resident.pop_front();
        }

        std::exception_ptr error = nullptr;
        const auto busy_from = std::chrono::steady_clock::now();
        auto finished_at = busy_from;

        try {

            if (!estimator) {

                estimator = makePathEstimator(model, params, engine);
            }

            // (the batch is this engine's from here on: freed below through this worker's context, not the uploader's — that one
            // is busy with the next copy)
            job->device_batch->finish(engine);
            finished_at = std::chrono::steady_clock::now();
            estimator->estimateBatchSeeded(job->estimates, *job->device_batch, params.rng_seed);

        } catch (...) {

            error = std::current_exception();
        }

        job->device_batch.reset();
        const auto done_at = std::chrono::steady_clock::now();

        std::vector<std::unique_ptr<Job> > dropped;  // (destroyed behind the block: without the mutex)

        {
            std::lock_guard<std::mutex> lock(mutex);

            busy_estimates.erase(job->estimates);
            --num_resident;
            --num_unfinished;
            wait_for_batch_seconds += std::chrono::duration<double>(busy_from - idle_from).count();
            finish_seconds += std::chrono::duration<double>(finished_at - busy_from).count();
            estimate_seconds += std::chrono::duration<double>(done_at - finished_at).count();
            completions.emplace_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - stats_epoch).count());

            if (error) {

                dropped = fail(error);
            }
        }

        changed.notify_all();
    }
}

void BatchPipeline::stats(rpvg_hip_kernel_stats * stats_out) const {

    std::vector<const HipEngine *> engines;

    for (auto & engine: worker_engines) {

        engines.emplace_back(engine.get());
    }

    HipEngine::stats(engines, stats_out);
}

void BatchPipeline::resetStats() const {

    for (auto & engine: worker_engines) {

        engine->resetStats();
    }

    for (auto & engine: uploader_engines) {

        engine->resetStats();
    }

    std::lock_guard<std::mutex> lock(mutex);
    BatchPipeline * self = const_cast<BatchPipeline *>(this);
    self->upload_seconds = 0;
    self->finish_seconds = 0;
    self->estimate_seconds = 0;
    self->wait_for_batch_seconds = 0;
    self->upload_batches = 0;
    self->stats_epoch = std::chrono::steady_clock::now();
    self->completions.clear();
}

void BatchPipeline::workerSeconds(double * finish_out, double * estimate_out, double * idle_out) const {

    std::lock_guard<std::mutex> lock(mutex);
    const double batches = std::max<double>(1, completions.size());

    *finish_out = finish_seconds / batches;
    *estimate_out = estimate_seconds / batches;
    *idle_out = wait_for_batch_seconds / batches;
}

void BatchPipeline::uploadDeviceMs(double * copies_ms_out, double * kernels_ms_out) const {

    *copies_ms_out = 0;
    *kernels_ms_out = 0;

    for (auto & engine: uploader_engines) {

        rpvg_hip_kernel_stats stats;
        engine->stats(&stats);

        *copies_ms_out += stats.h2d_ms;
        *kernels_ms_out += stats.build_ms;
    }
}

std::vector<double> BatchPipeline::completionSeconds() const {

    std::lock_guard<std::mutex> lock(mutex);
    return completions;
}

double BatchPipeline::meanUploadSeconds(uint64_t * batches_out) const {

    std::lock_guard<std::mutex> lock(mutex);

    if (batches_out) {

        *batches_out = upload_batches;
    }

    return upload_batches ? upload_seconds / upload_batches : 0.0;
}

}
