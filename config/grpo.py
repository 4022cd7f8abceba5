"""Entry point kept from the reference: `--config config/grpo.py:<experiment>` (config/grpo.py:432-433 upstream).
The experiment tables live in adv_grpo_amd/config/experiments.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adv_grpo_amd.config.experiments import EXPERIMENTS, compressibility  # noqa: E402,F401
from adv_grpo_amd.config.experiments import get_config as _get  # noqa: E402


def get_config(name, gpu_number=8):
    return _get(name, gpu_number)


# the reference exposes one function per experiment; keep that surface too
def _make(n):
    return lambda gpu_number=8: _get(n, gpu_number)


for _n in EXPERIMENTS:
    globals()[_n] = _make(_n)
