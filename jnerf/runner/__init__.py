from jnerf_amd.runner import Runner  # noqa: F401
from jnerf_amd.neus_runner import NeuSRunner  # noqa: F401
