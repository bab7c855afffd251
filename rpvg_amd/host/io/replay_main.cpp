// rpvg_amd_replay — command-line front end of replayInference(): the inference stage of rpvg on a
// dumped input.  Option letters follow rpvg's (src/main.cpp:364-419) where they exist.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "replay.hpp"

using namespace rpvg_amd;

static void usage() {

    std::cerr << "usage: rpvg_amd_replay -p <probs.txt[.gz]> -o <output-prefix> [-f <path-info.tsv[.gz]>]\n"
                 "         [-i transcripts|strains|haplotype-transcripts|haplotypes] [-y ploidy] [-n gibbs-samples]\n"
                 "         [--max-em-its N] [--max-rel-em-conv X] [--gibbs-thin-its N] [--min-hap-prob X]\n"
                 "         [--prob-precision X] [--ind-hap-inference] [--use-hap-gibbs] [-r seed] [-d device]\n"
                 "         [--unaligned-reads N]" << std::endl;
}

int main(int argc, char ** argv) {

    std::string probs_filename, path_info_filename, output_prefix, inference_model = "haplotype-transcripts";
    rpvg_params params = rpvg_params_default();
    int device = 0;
    uint32_t unaligned_read_count = 0;

    for (int i = 1; i < argc; ++i) {

        const std::string arg = argv[i];
        auto value = [&]() -> const char * { if (i + 1 >= argc) { usage(); std::exit(1); } return argv[++i]; };

        if (arg == "-p") probs_filename = value();
        else if (arg == "-f") path_info_filename = value();
        else if (arg == "-o") output_prefix = value();
        else if (arg == "-i") inference_model = value();
        else if (arg == "-y") params.ploidy = std::atoi(value());
        else if (arg == "-n") params.num_gibbs_samples = std::atoi(value());
        else if (arg == "-r") params.rng_seed = std::strtoul(value(), nullptr, 10);
        else if (arg == "-d") device = std::atoi(value());
        else if (arg == "--max-em-its") params.max_em_its = std::atoi(value());
        else if (arg == "--max-rel-em-conv") params.max_rel_em_conv = std::atof(value());
        else if (arg == "--gibbs-thin-its") params.gibbs_thin_its = std::atoi(value());
        else if (arg == "--min-hap-prob") params.min_hap_prob = std::atof(value());
        else if (arg == "--prob-precision") params.prob_precision = std::atof(value());
        else if (arg == "--ind-hap-inference") params.ind_hap_inference = 1;
        else if (arg == "--use-hap-gibbs") params.use_hap_gibbs = 1;
        else if (arg == "--unaligned-reads") unaligned_read_count = std::strtoul(value(), nullptr, 10);
        else { usage(); return 1; }
    }

    if (probs_filename.empty() || output_prefix.empty()) {

        usage();
        return 1;
    }

    try {

        const size_t num_clusters = replayInference(probs_filename, path_info_filename, inference_model, params, output_prefix, device, unaligned_read_count);
        std::cerr << "Inferred " << inference_model << " estimates for " << num_clusters << " clusters" << std::endl;

    } catch (const std::exception & e) {

        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }

    return 0;
}
