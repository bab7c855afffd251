#include "cluster_io.hpp"

#include <algorithm>
#include <cmath>
#include <iomanip>
#include <sstream>
#include <stdexcept>

#include <zlib.h>

namespace rpvg_amd {

namespace {

// src/threaded_output_writer.cpp:6
const uint32_t out_precision_digits = 8;

}

// Whole (possibly gzipped) file as text; gzread passes plain files through.
std::string readTextFile(const std::string & filename) {

    gzFile file = gzopen(filename.c_str(), "rb");

    if (!file) {

        throw std::runtime_error("cannot open " + filename);
    }

    std::string text;
    std::vector<char> buffer(1 << 20);

    while (true) {

        const int read_size = gzread(file, buffer.data(), buffer.size());

        if (read_size < 0) {

            gzclose(file);
            throw std::runtime_error("cannot read " + filename);
        }

        if (read_size == 0) {

            break;
        }

        text.append(buffer.data(), read_size);
    }

    gzclose(file);
    return text;
}

void writeTextFile(const std::string & filename, const std::string & text) {

    const bool gzip = filename.size() > 3 && filename.compare(filename.size() - 3, 3, ".gz") == 0;

    // "T" = transparent: zlib then writes the bytes as they are
    gzFile file = gzopen(filename.c_str(), gzip ? "wb" : "wbT");

    if (!file) {

        throw std::runtime_error("cannot open " + filename + " for writing");
    }

    size_t written = 0;

    while (written < text.size()) {

        const int chunk = gzwrite(file, text.data() + written, std::min<size_t>(text.size() - written, 1 << 30));

        if (chunk <= 0) {

            gzclose(file);
            throw std::runtime_error("cannot write " + filename);
        }

        written += chunk;
    }

    gzclose(file);
}

namespace {

std::vector<std::string> splitString(const std::string & text, const char delim) {

    std::vector<std::string> elems;
    std::stringstream ss(text);
    std::string item;

    while (std::getline(ss, item, delim)) {

        elems.emplace_back(item);
    }

    return elems;
}

}

std::vector<ProbabilityCluster> readProbabilityClusters(const std::string & filename, const double prob_precision) {

    std::vector<ProbabilityCluster> clusters;

    std::stringstream text(readTextFile(filename));
    std::string line;

    bool expect_paths = false;

    while (std::getline(text, line)) {

        if (line.empty()) {

            continue;
        }

        if (line == "#" || line.compare(0, 2, "# ") == 0) {

            clusters.emplace_back(ProbabilityCluster());
            expect_paths = true;

            if (line.size() > 2) {

                // "# <alignment-path lists> <cluster index>": the rank key of src/main.cpp:811-827 (an extension of this
                // project's writer; the reference writes a bare "#")
                const auto fields = splitString(line.substr(2), ' ');

                if (fields.size() != 2) {

                    throw std::runtime_error(filename + ": cluster marker \"" + line + "\" is neither \"#\" nor \"# <lists> <index>\"");
                }

                // (digits only: std::stoull would take "12abc", and its own exceptions name neither the file nor the line)
                for (auto & field: fields) {

                    if (field.empty() || field.size() > 19 || field.find_first_not_of("0123456789") != std::string::npos) {

                        throw std::runtime_error(filename + ": cluster marker \"" + line + "\": \"" + field + "\" is not a number");
                    }
                }

                clusters.back().has_rank_key = true;
                clusters.back().num_align_lists = std::stoull(fields.at(0));
                clusters.back().cluster_index = std::stoull(fields.at(1));
            }

            continue;
        }

        if (clusters.empty()) {

            throw std::runtime_error(filename + ": data before the first cluster marker");
        }

        if (expect_paths) {

            // <name>,<length>,<effective_length> ...
            for (auto & field: splitString(line, ' ')) {

                const size_t second_comma = field.rfind(',');
                const size_t first_comma = (second_comma == std::string::npos) ? std::string::npos : field.rfind(',', second_comma - 1);

                if (first_comma == std::string::npos) {

                    throw std::runtime_error(filename + ": malformed path field '" + field + "'");
                }

                PathInfo path(field.substr(0, first_comma));
                path.length = std::stoul(field.substr(first_comma + 1, second_comma - first_comma - 1));
                path.effective_length = std::stod(field.substr(second_comma + 1));

                clusters.back().paths.emplace_back(std::move(path));
            }

            expect_paths = false;
            continue;
        }

        // <read_count> <noise_prob> [<prob>:<idx>[,<idx>]...]...
        const auto fields = splitString(line, ' ');

        if (fields.size() < 2) {

            throw std::runtime_error(filename + ": malformed row '" + line + "'");
        }

        ReadPathProbabilities::PathProbs path_probs;

        for (size_t i = 2; i < fields.size(); ++i) {

            const size_t colon = fields[i].find(':');

            if (colon == std::string::npos) {

                throw std::runtime_error(filename + ": malformed probability group '" + fields[i] + "'");
            }

            std::vector<uint32_t> path_ids;

            for (auto & idx: splitString(fields[i].substr(colon + 1), ',')) {

                path_ids.emplace_back(std::stoul(idx));

                if (path_ids.back() >= clusters.back().paths.size()) {

                    throw std::runtime_error(filename + ": path index out of range in '" + line + "'");
                }
            }

            path_probs.emplace_back(std::stod(fields[i].substr(0, colon)), std::move(path_ids));
        }

        clusters.back().cluster_probs.emplace_back(std::stoul(fields[0]), std::stod(fields[1]), path_probs, prob_precision);
    }

    return clusters;
}

void writeProbabilityClusters(const std::string & filename, const std::vector<ProbabilityCluster> & clusters, const double prob_precision) {

    // digits as the reference chooses them (src/threaded_output_writer.cpp:40)
    const uint32_t prob_precision_digits = std::max(out_precision_digits, static_cast<uint32_t>(std::ceil(-1 * std::log10(prob_precision))));

    std::stringstream out;

    for (auto & cluster: clusters) {

        if (cluster.cluster_probs.empty() || cluster.paths.empty()) {

            continue;
        }

        if (cluster.has_rank_key) {

            out << "# " << cluster.num_align_lists << " " << cluster.cluster_index << std::endl;

        } else {

            out << "#" << std::endl;
        }

        out << std::setprecision(out_precision_digits);

        for (size_t i = 0; i < cluster.paths.size(); ++i) {

            out << (i ? " " : "") << cluster.paths[i].name << "," << cluster.paths[i].length << "," << cluster.paths[i].effective_length;
        }

        out << std::endl;
        out << std::setprecision(prob_precision_digits);

        for (auto & read_path_probs: cluster.cluster_probs) {

            out << read_path_probs.readCount() << " " << read_path_probs.noiseProb();

            for (auto & path_probs: read_path_probs.pathProbs()) {

                out << " " << path_probs.first << ":";

                for (size_t i = 0; i < path_probs.second.size(); ++i) {

                    out << (i ? "," : "") << path_probs.second[i];
                }
            }

            out << std::endl;
        }
    }

    writeTextFile(filename, out.str());
}

std::unordered_map<std::string, PathInfo> parseHaplotypeTranscriptInfo(const std::string & filename, const bool parse_haplotype_ids, const bool use_transcript_names) {

    std::unordered_map<std::string, PathInfo> haplotype_transcript_info;

    std::unordered_map<std::string, uint32_t> transcript_id_index;
    std::unordered_map<std::string, uint32_t> haplotype_id_index;

    std::stringstream text(readTextFile(filename));
    std::string line;

    bool is_first_line = true;
    bool is_old_format = false;

    while (std::getline(text, line)) {

        if (line.empty()) {

            continue;
        }

        const auto fields = splitString(line, '\t');

        if (is_first_line) {

            if (fields.empty() || fields.front() != "Name") {

                throw std::runtime_error(filename + ": header does not start with 'Name'");
            }

            // the old five-column format carries a Reference column before Haplotypes (src/main.cpp:296-299)
            is_old_format = (line.find("Reference") != std::string::npos);
            is_first_line = false;
            continue;
        }

        const size_t num_columns = is_old_format ? 5 : 4;

        if (fields.size() < num_columns) {

            throw std::runtime_error(filename + ": too few columns in '" + line + "'");
        }

        auto info_it = haplotype_transcript_info.emplace(fields[0], PathInfo(fields[0]));

        if (!info_it.second) {

            throw std::runtime_error(filename + ": path '" + fields[0] + "' is listed twice");
        }

        PathInfo & info = info_it.first->second;
        const std::string & transcript = fields[2];

        if (use_transcript_names) {

            info.name = transcript;
        }

        info.group_id = transcript_id_index.emplace(transcript, transcript_id_index.size()).first->second;

        const std::string & haplotypes = fields[num_columns - 1];

        if (parse_haplotype_ids) {

            for (auto & haplotype: splitString(haplotypes, ',')) {

                info.source_ids.emplace(haplotype_id_index.emplace(haplotype, haplotype_id_index.size()).first->second);
            }

            info.source_count = info.source_ids.size();

        } else {

            info.source_count = std::count(haplotypes.begin(), haplotypes.end(), ',') + 1;
        }
    }

    return haplotype_transcript_info;
}

void applyHaplotypeTranscriptInfo(std::vector<ProbabilityCluster> * clusters, const std::unordered_map<std::string, PathInfo> & haplotype_transcript_info) {

    for (auto & cluster: *clusters) {

        for (auto & path: cluster.paths) {

            auto info_it = haplotype_transcript_info.find(path.name);

            if (info_it == haplotype_transcript_info.end()) {

                throw std::runtime_error("path '" + path.name + "' is missing from the path info file");
            }

            path.group_id = info_it->second.group_id;
            path.source_count = info_it->second.source_count;
            path.source_ids = info_it->second.source_ids;
        }
    }
}

}
