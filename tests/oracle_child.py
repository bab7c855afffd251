"""The CPU oracle in a process of its own (test infrastructure): reads a pickled (model, parameter keywords, batch, threads) from
stdin, writes the pickled estimates to stdout.  The restatement keeps the reference's assertions, and with thresholds far from the
defaults (min_hap_prob 1e-5: tens of thousands of subset weights) the reference's own `sum_hap_prob <= 1`
(src/path_abundance_estimator.cpp:748) can fail on rounding: an abort here ends this process, not a sweep."""
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from oracle import pyoracle
    from rpvg_amd.batch import make_params
    model, kw, batch, threads = pickle.load(sys.stdin.buffer)
    ref, _ = pyoracle.run(model, make_params(**kw), batch, threads)
    pickle.dump(ref, sys.stdout.buffer, protocol=pickle.HIGHEST_PROTOCOL)
    sys.stdout.buffer.flush()


if __name__ == "__main__":
    main()
