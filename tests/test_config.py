"""CPU: the kept `config/grpo.py:<name>` entry point and the fields the trainers read."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entry_point_and_field_values():
    from adv_grpo_amd.config.experiments import parse_config_flag
    cfg = parse_config_flag(os.path.join(ROOT, "config", "grpo.py") + ":pickscore_cotrain_sd3_fast")
    # values of config/grpo.py:315-376 upstream
    assert cfg.sample.num_steps == 10 and cfg.sample.train_num_steps == 2 and cfg.sample.guidance_scale == 4.5
    assert cfg.sample.num_image_per_prompt == 16 and cfg.sample.mini_num_image_per_prompt == 8
    assert cfg.sample.num_batches_per_epoch == 12 and cfg.train.gradient_accumulation_steps == 6
    assert cfg.train.clip_range == 1e-5 and cfg.sample.noise_level == 0.8 and cfg.sample.global_std is True
    assert cfg.d_times == 20 and cfg.d_lr == 5e-6 and cfg.tune_layer == -1 and cfg.train_d is True
    assert cfg.reward_fn.to_dict() == {"pickscore_cotrain": 1} and cfg.resolution == 512
    assert cfg.train.learning_rate == 3e-4 and cfg.train.adam_weight_decay == 1e-4 and cfg.train.max_grad_norm == 1.0
    assert cfg.mixed_precision == "bf16" and cfg.sample.random_timestep == 0 and cfg.train.ema is True
    d = parse_config_flag(os.path.join(ROOT, "config", "grpo.py") + ":dino_cotrain_sd3_patch_fast")
    assert d.reward_fn.to_dict() == {"dino_patch_cotrain": 1} and d.d_times == 10 and d.d_lr == 1e-4
    m = parse_config_flag(os.path.join(ROOT, "config", "grpo.py") + ":pickscore_sd3_fast")
    assert m.reward_fn.to_dict() == {"pickscore": 0.5, "ocr": 0.5} and m.sample.random_timestep is None
    assert "train_d" not in m
    with pytest.raises(AttributeError):
        cfg.no_such_field
    one = parse_config_flag(os.path.join(ROOT, "config", "grpo.py") + ":eval_sd3_fast", gpu_number=1)
    assert one.sample.num_batches_per_epoch == 48     # k = 1 on one GPU: 48 prompt groups per epoch
