"""adv_grpo_amd -- MI355X-native Adv-GRPO SD3 rollout-and-update hot path.

Host-side mirror of the reference's operator surface (same module / function names as
``adv_grpo.*`` for the path SURVEY.md section 8 scopes) over hand-written gfx950 kernels in
``libadvgrpo_hip.so`` (C ABI: include/advgrpo.h).  No CPU fallback exists.
"""
__version__ = "0.1.0"

import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in the order of their first use; this path keeps up to
# seven streams alive (main, adapter gradients, two rollouts, their VAE side streams, reward scoring), and two streams on one
# queue do not overlap at all (ops.concurrent_stream).  Only effective when set before the HIP runtime initialises -- bench.py,
# scripts/train_sd3_fast.py and tests/conftest.py set it first thing; this line covers every other importer that gets here early.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
