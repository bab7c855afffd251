// Several cluster batches in flight on one GPU.
//
// The reference estimates its clusters from an OpenMP team, every thread a cluster of its own at any time
// (src/main.cpp:829-998: independent clusters, no barrier between them).  On the GPU the unit is a batch of clusters, and
// one batch is a chain of dependent kernels — matrices, their row collapse, the diploid search, the path subsets, the EM —
// whose tail (a few wavefronts iterating the slowest EM problems) leaves most of the GPU idle.  The pipeline keeps several
// batches at different stages of that chain on the GPU at once:
//   submit()      hands over the host arrays of a batch (the caller's, valid until the batch is done) and the containers its
//                 estimates go to;
//   one uploader  thread with a context of its own (rpvg_hip_create_uploader: the device's highest stream priority) copies
//                 the rows and the path side of batch after batch — at most `workers + 3` batches are on the GPU; nothing but
//                 the copies (rpvg_hip_batch_upload_begin);
//   one more      thread queues the kernels behind a batch's copies — offsets from their counts, expansion, validation, read
//                 totals, haplotype columns — on the uploader's side stream (rpvg_hip_batch_upload_finish_queue: no context, no
//                 hardware queue more) while the uploader copies the next batch;
//   `workers`     estimator threads, each with a single-lane engine (its main stream on a hardware queue of its own, three
//                 side streams: rpvg_hip_create_with_streams) and an estimator of its own, take the batches in order, wait for
//                 those kernels (rpvg_hip_batch_upload_finish_wait) and run PathEstimator::estimateBatchSeeded.
// configs[2] of BASELINE.json on one MI355X: 7 ms per resident batch one at a time (two host lanes), 4.1-4.5 through the
// pipeline with every batch copied inside the clock (190 MB per batch at 53 GB/s: 3.6 ms — the PCIe link is the next bound);
// how it got there: docs/design/history-r05.md.
// Results are those of the same calls made one after the other: batches do not interact.
#ifndef RPVG_AMD_BATCH_PIPELINE_HPP
#define RPVG_AMD_BATCH_PIPELINE_HPP

#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rpvg_batch.h"
#include "hip_engine.hpp"
#include "path_cluster_estimates.hpp"
#include "path_estimator.hpp"

namespace rpvg_amd {

class BatchPipeline {

    public:

        // workers: batches estimated side by side; 0 = the default (RPVG_AMD_PIPELINE_WORKERS, else 4 — 6 with --use-hap-gibbs)
        BatchPipeline(const int device, const std::string & model, const rpvg_params & params, const int workers = 0);
        ~BatchPipeline();

        BatchPipeline(const BatchPipeline &) = delete;
        BatchPipeline & operator=(const BatchPipeline &) = delete;

        // Queues a batch.  host_batch's arrays and *estimates (one entry per cluster, PathInfo filled in as
        // src/main.cpp:855-887 does) belong to the pipeline until the batch is done; a second batch on the same
        // containers waits for the first.  Cluster i draws from mt19937(params.rng_seed + i) (src/main.cpp:976).
        void submit(const rpvg_cluster_batch & host_batch, std::vector<PathClusterEstimates> * estimates);

        // Returns when every submitted batch is done; rethrows the first failure (the batches behind it are dropped).
        void wait();

        int numWorkers() const { return worker_threads.size(); }

        // Kernel statistics of all workers together (include/rpvg_hip.h), and a reset.
        void stats(rpvg_hip_kernel_stats * stats_out) const;
        void resetStats() const;

        // wall seconds the uploader spent per batch since the last reset (mean), and their number
        double meanUploadSeconds(uint64_t * batches_out = nullptr) const;

        // mean wall seconds per batch a worker spent finishing the upload, estimating, and waiting for a resident batch
        void workerSeconds(double * finish_out, double * estimate_out, double * idle_out) const;

        // device time of the uploads since the last reset: the copies (HIP-event span around them) and the kernels behind them
        void uploadDeviceMs(double * copies_ms_out, double * kernels_ms_out) const;

        // when the batches since the last reset were done, in seconds since that reset, in order of completion
        std::vector<double> completionSeconds() const;

    private:

        struct Job {

            rpvg_cluster_batch host_batch;
            std::vector<PathClusterEstimates> * estimates;
            std::unique_ptr<DeviceClusterBatch> device_batch;
        };

        void uploadLoop(const int uploader);
        void queueLoop();
        void workerLoop(const int worker);
        // (mutex held) records the error and takes the batches that have not started out of the queues: the counters are settled
        // here, the jobs — device batches whose destruction waits for their copies — are the caller's to destroy once it has let
        // go of the mutex
        std::vector<std::unique_ptr<Job> > fail(std::exception_ptr error);

        const int device;
        const std::string model;
        const rpvg_params params;

        std::vector<std::shared_ptr<HipEngine> > uploader_engines;
        std::vector<std::shared_ptr<HipEngine> > worker_engines;

        mutable std::mutex mutex;
        std::condition_variable changed;

        std::deque<std::unique_ptr<Job> > to_upload;
        std::deque<std::unique_ptr<Job> > copied;                         // copied, the kernels behind the copies not yet queued
        std::deque<std::unique_ptr<Job> > resident;                       // uploaded, waiting for a worker
        std::set<const void *> busy_estimates;

        size_t num_resident;   // uploaded or being uploaded, not yet retired
        size_t num_unfinished;
        bool stopping;
        std::exception_ptr first_error;

        double upload_seconds, finish_seconds, estimate_seconds, wait_for_batch_seconds;
        uint64_t upload_batches;
        std::chrono::steady_clock::time_point stats_epoch;
        std::vector<double> completions;

        std::vector<std::thread> upload_threads;
        std::thread queue_thread;
        size_t max_resident;
        std::vector<std::thread> worker_threads;
};

}

#endif
