#ifndef RPVG_AMD_EXPERIMENTS_HPP
#define RPVG_AMD_EXPERIMENTS_HPP

#include <cstdlib>

// A/B switches of the host layer's measurements (docs/design/knobs.md): read only by a build with -DRPVG_AMD_EXPERIMENTS
// (`make -C rpvg_amd/host clean all EXPERIMENTS=1`); the shipped library does not look at them.
#ifdef RPVG_AMD_EXPERIMENTS
#define RPVG_AMD_EXPERIMENT_ENV(name) std::getenv(name)
#else
#define RPVG_AMD_EXPERIMENT_ENV(name) (static_cast<const char *>(nullptr))
#endif

#endif
