#include "estimator_factory.hpp"

#include "path_abundance_estimator.hpp"
#include "path_posterior_estimator.hpp"

namespace rpvg_amd {

std::unique_ptr<PathEstimator> makePathEstimator(const std::string & inference_model, const rpvg_params & params, std::shared_ptr<HipEngine> engine) {

    if (inference_model == "haplotypes") {

        return std::unique_ptr<PathEstimator>(new PathGroupPosteriorEstimator(params.ploidy, params.use_hap_gibbs, params.prob_precision, engine));

    } else if (inference_model == "transcripts") {

        return std::unique_ptr<PathEstimator>(new PathAbundanceEstimator(params.max_em_its, params.max_rel_em_conv, params.num_gibbs_samples, params.gibbs_thin_its, params.prob_precision, engine));

    } else if (inference_model == "haplotype-transcripts") {

        return std::unique_ptr<PathEstimator>(new NestedPathAbundanceEstimator(params.ploidy, params.min_hap_prob, !params.ind_hap_inference, params.use_hap_gibbs, params.max_em_its, params.max_rel_em_conv, params.num_gibbs_samples, params.gibbs_thin_its, params.prob_precision, engine));

    } else if (inference_model == "strains") {

        return std::unique_ptr<PathEstimator>(new MinimumPathAbundanceEstimator(params.max_em_its, params.max_rel_em_conv, params.num_gibbs_samples, params.gibbs_thin_its, params.prob_precision, engine));
    }

    throw EngineError("unknown inference model '" + inference_model + "'");
}

}
