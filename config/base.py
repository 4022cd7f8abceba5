"""Defaults (upstream config/base.py:4-113) -- see adv_grpo_amd/config/experiments.py:DEFAULTS."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd.config.experiments import base_config as get_config  # noqa: E402,F401
