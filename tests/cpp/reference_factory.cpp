// The estimator factory of the reference's caller (src/main.cpp:766-788) with nothing changed but the namespace:
// the constructors take exactly the reference's parameter lists (src/path_abundance_estimator.hpp:22,55,
// src/path_posterior_estimator.hpp:33) and bind to the process-default engine.  Compiled by tests/test_abi_and_host.py
// (no GPU needed to compile and link); run with a model name on a GPU box it estimates one small cluster through the
// reference-shaped estimate() call (src/main.cpp:976-977) and prints the abundances.
#include <cassert>
#include <cstdlib>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "path_abundance_estimator.hpp"
#include "path_posterior_estimator.hpp"

using namespace rpvg_amd;
using namespace std;

int main(int argc, char * argv[]) {

    const string inference_model = argc > 1 ? argv[1] : "";

    const uint32_t ploidy = 2;
    const bool use_hap_gibbs = false;
    const bool ind_hap_inference = false;
    const double min_hap_prob = 0.001;
    const uint32_t max_em_its = 10000;
    const double max_rel_em_conv = 0.001;
    const uint32_t num_gibbs_samples = 0;
    const uint32_t gibbs_thin_its = 25;
    const double prob_precision = 1e-8;

    if (inference_model.empty()) {

        cout << "usage: reference_factory <haplotypes|transcripts|strains|haplotype-transcripts>" << endl;
        return 0;
    }

    PathEstimator * path_estimator;

    if (inference_model == "haplotypes") {

        path_estimator = new PathGroupPosteriorEstimator(ploidy, use_hap_gibbs, prob_precision);

    } else if (inference_model == "transcripts") {

        path_estimator = new PathAbundanceEstimator(max_em_its, max_rel_em_conv, num_gibbs_samples, gibbs_thin_its, prob_precision);

    } else if (inference_model == "strains") {

        path_estimator = new MinimumPathAbundanceEstimator(max_em_its, max_rel_em_conv, num_gibbs_samples, gibbs_thin_its, prob_precision);

    } else if (inference_model == "haplotype-transcripts") {

        path_estimator = new NestedPathAbundanceEstimator(ploidy, min_hap_prob, !ind_hap_inference, use_hap_gibbs, max_em_its, max_rel_em_conv, num_gibbs_samples, gibbs_thin_its, prob_precision);

    } else {

        assert(false);
        return 1;
    }

    // KAT-EM-disjoint of SURVEY.md §8c: two paths, 30 reads on the first and 70 on the second, noise 1e-4
    PathClusterEstimates estimates;
    estimates.paths.emplace_back(PathInfo("a"));
    estimates.paths.emplace_back(PathInfo("b"));
    estimates.paths.at(0).group_id = 0;
    estimates.paths.at(1).group_id = 0;
    estimates.paths.at(0).source_ids = {0};
    estimates.paths.at(1).source_ids = {1};

    vector<ReadPathProbabilities> cluster_probs;
    cluster_probs.emplace_back(30, 1e-4, ReadPathProbabilities::PathProbs({{1 - 1e-4, {0}}}), prob_precision);
    cluster_probs.emplace_back(70, 1e-4, ReadPathProbabilities::PathProbs({{1 - 1e-4, {1}}}), prob_precision);

    mt19937 mt_rng(0);
    path_estimator->estimate(&estimates, cluster_probs, &mt_rng);
    if (getenv("RPVG_AMD_TRACE_EXIT")) cerr << "[exit] estimate() returned" << endl;

    cout << inference_model << " total " << estimates.total_count << " noise " << estimates.noise_count << " abundances";

    for (auto & abundance: estimates.abundances) {

        cout << " " << abundance;
    }

    cout << " posteriors";

    for (auto & posterior: estimates.posteriors) {

        cout << " " << posterior;
    }

    cout << endl;

    delete path_estimator;
    return 0;
}
