// Result files on the output side of the inference hot path, in the formats of the reference's
// writers (src/threaded_output_writer.{hpp,cpp}): tab-separated, std::setprecision(8) default float
// formatting, one `Unknown` row carrying the noise read count.
//
//   AbundanceEstimatesWriter                 <prefix>.txt        -i transcripts / strains      :283-343
//   HaplotypeAbundanceEstimatesWriter        <prefix>.txt        -i haplotype-transcripts      :346-432
//   JointHaplotypeAbundanceEstimatesWriter   <prefix>_joint.txt  -i haplotype-transcripts      :434-546
//   JointHaplotypeEstimatesWriter            <prefix>.txt        -i haplotypes                 :233-280
//   ReadCountGibbsSamplesWriter              <prefix>_gibbs.txt.gz  -n > 0                     :98-230
//
// The reference streams string buffers to a BGZF writer thread; these writers collect the text and write
// the file on close() (plain text, gzip for the ".gz" one).  Row order across clusters follows the order
// of the addEstimates() calls (the reference's order depends on its OpenMP schedule).
#ifndef RPVG_AMD_ESTIMATES_WRITERS_HPP
#define RPVG_AMD_ESTIMATES_WRITERS_HPP

#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../path_cluster_estimates.hpp"

namespace rpvg_amd {

typedef std::vector<std::pair<uint32_t, PathClusterEstimates> > ClusterEstimatesList;

// Sum of abundance / effective length over every group-set member of every cluster
// (src/main.cpp:1029-1057): the TPM denominator.
double totalTranscriptCount(const ClusterEstimatesList & path_cluster_estimates);

class EstimatesWriter {

    public:

        explicit EstimatesWriter(const std::string & filename_in);
        virtual ~EstimatesWriter() {};

        void close();

    protected:

        std::stringstream out;

    private:

        const std::string filename;
};

class AbundanceEstimatesWriter : public EstimatesWriter {

    public:

        AbundanceEstimatesWriter(const std::string filename_prefix, const double total_transcript_count_in);

        void addEstimates(const ClusterEstimatesList & path_cluster_estimates);
        void addNoiseTranscript(const uint32_t unaligned_read_count);

    private:

        const double total_transcript_count;
        double noise_count;
};

class HaplotypeAbundanceEstimatesWriter : public EstimatesWriter {

    public:

        HaplotypeAbundanceEstimatesWriter(const std::string filename_prefix, const uint32_t ploidy_in, const double total_transcript_count_in);

        void addEstimates(const ClusterEstimatesList & path_cluster_estimates);
        void addNoiseTranscript(const uint32_t unaligned_read_count);

    private:

        const uint32_t ploidy;
        const double total_transcript_count;
        double noise_count;
};

class JointHaplotypeAbundanceEstimatesWriter : public EstimatesWriter {

    public:

        JointHaplotypeAbundanceEstimatesWriter(const std::string filename_prefix, const uint32_t ploidy_in, const double min_posterior_in, const double total_transcript_count_in);

        void addEstimates(const ClusterEstimatesList & path_cluster_estimates);
        void addNoiseTranscript(const uint32_t unaligned_read_count);

    private:

        const uint32_t ploidy;
        const double min_posterior;
        const double total_transcript_count;
        std::vector<double> noise_counts;
};

class JointHaplotypeEstimatesWriter : public EstimatesWriter {

    public:

        JointHaplotypeEstimatesWriter(const std::string filename_prefix, const uint32_t ploidy_in, const double min_posterior_in);

        void addEstimates(const ClusterEstimatesList & path_cluster_estimates);

    private:

        const uint32_t ploidy;
        const double min_posterior;
};

class ReadCountGibbsSamplesWriter : public EstimatesWriter {

    public:

        ReadCountGibbsSamplesWriter(const std::string filename_prefix, const uint32_t num_gibbs_samples_in);

        void addSamples(const std::pair<uint32_t, PathClusterEstimates> & path_cluster_estimate);
        void addNoiseTranscript(const uint32_t unaligned_read_count);

    private:

        const uint32_t num_gibbs_samples;
        std::vector<double> noise_counts;
};

}

#endif
