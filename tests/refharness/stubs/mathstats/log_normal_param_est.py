def GapEstimator(mu, sigma, read_len, samples, c1_len, c2_len=None):
    raise NotImplementedError('log-normal gap estimation is outside the parity scope (SURVEY.md App. C.1)')
