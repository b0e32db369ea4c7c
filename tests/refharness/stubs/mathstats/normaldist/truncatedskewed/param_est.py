from besst_amd.mathstats_compat import GapEstimator, tr_sk_std_dev, PreCalcMLvaluesOfdLongContigs  # noqa: F401
