"""`jnerf` import alias: scripts written against the reference's package layout (python/jnerf: `from jnerf.runner import Runner`,
`from jnerf.utils.config import init_cfg, get_cfg`, `from jnerf.utils.registry import build_from_cfg, NETWORKS, ...`) resolve to jnerf_amd, the MI355X path.
Nothing lives here but re-exports."""
from jnerf_amd import __version__  # noqa: F401
