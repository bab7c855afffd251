// Estimator selection by inference model name, as the reference's driver does
// (src/main.cpp:766-788), for the GPU-backed classes.
#ifndef RPVG_AMD_ESTIMATOR_FACTORY_HPP
#define RPVG_AMD_ESTIMATOR_FACTORY_HPP

#include <memory>
#include <string>

#include "../../include/rpvg_batch.h"
#include "path_estimator.hpp"

namespace rpvg_amd {

// inference_model: "haplotypes" | "transcripts" | "strains" | "haplotype-transcripts".
std::unique_ptr<PathEstimator> makePathEstimator(const std::string & inference_model, const rpvg_params & params, std::shared_ptr<HipEngine> engine);

}

#endif
