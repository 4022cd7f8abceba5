from .config_dict import ConfigDict  # noqa: F401
from .experiments import get_config, base_config, EXPERIMENTS  # noqa: F401
