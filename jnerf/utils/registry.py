from jnerf_amd.utils.registry import *  # noqa: F401,F403
from jnerf_amd.utils.registry import Registry, build_from_cfg, NETWORKS, ENCODERS, DATASETS, OPTIMS, SAMPLERS, LOSSES, SCHEDULERS  # noqa: F401
