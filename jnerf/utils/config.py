from jnerf_amd.utils.config import *  # noqa: F401,F403
from jnerf_amd.utils.config import Config, init_cfg, get_cfg, update_cfg, save_cfg  # noqa: F401
