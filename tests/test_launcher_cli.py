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
