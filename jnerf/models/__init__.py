from jnerf_amd import encoders, network, networks_ori, sampler, losses  # noqa: F401  (registers the modules under the reference's registry names)
