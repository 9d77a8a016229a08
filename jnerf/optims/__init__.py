from jnerf_amd.optim import Adam, ExpDecay, EMA  # noqa: F401
