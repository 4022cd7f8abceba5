"""adv_grpo_amd -- MI355X-native Adv-GRPO SD3 rollout-and-update hot path.

Host-side mirror of the reference's operator surface (same module / function names as
``adv_grpo.*`` for the path SURVEY.md section 8 scopes) over hand-written gfx950 kernels in
``libadvgrpo_hip.so`` (C ABI: include/advgrpo.h).  No CPU fallback exists.
"""
__version__ = "0.1.0"
