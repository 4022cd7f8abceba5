"""The kept entry point's command line (scripts/train_sd3_fast.py, the stand-in for scripts/train_sd3_fast_*.py upstream): it parses
without a GPU and names the model switch by which the reference selects BASELINE configs 4 and 5 (config/grpo.py:324,330)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_help_lists_the_model_switch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_sd3_fast.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for needle in ("--model", "Qwen/Qwen-Image", "stable-diffusion-3.5-large", "--resolution", "--linear-dtype", "--config"):
        assert needle in r.stdout, needle


def test_launcher_refuses_the_unbuilt_multi_layer_dino_experiment_with_the_reason():
    """config/grpo.py ships `dino_cotrain_sd3_multi_fast`; its reward `dino_multi_cotrain` (RW:1032) is outside SURVEY.md 8's six scorers.  The
    config still parses (the table is the reference's), the launcher stops before touching a device and says why, and so does multi_score."""
    import pytest
    from adv_grpo_amd import rewards
    from adv_grpo_amd.config.experiments import get_config
    assert "dino_multi_cotrain" in get_config("dino_cotrain_sd3_multi_fast").reward_fn
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_sd3_fast.py"), "--config",
                        os.path.join(ROOT, "config", "grpo.py") + ":dino_cotrain_sd3_multi_fast"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "dino_multi_cotrain" in r.stderr and "not built" in r.stderr, r.stderr[-2000:]
    with pytest.raises(KeyError, match="not built"):
        rewards.multi_score("cpu", {"dino_multi_cotrain": 1.0})
