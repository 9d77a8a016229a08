from jnerf_amd.runner import Runner  # noqa: F401


class NeuSRunner:
    """projects/neus (SDF network + DTU data) is outside the Instant-NGP hot path this build covers (DESIGN.md 'out of scope')"""

    def __init__(self, *a, **k):
        raise NotImplementedError(self.__doc__)
