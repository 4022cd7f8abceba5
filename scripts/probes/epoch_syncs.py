"""Device -> host synchronisations inside one full epoch of the trainer (torch ops only: torch.cuda.set_sync_debug_mode), by call
site.  Each one stalls the host until the GPU has drained."""
import collections, os, sys, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adv_grpo_amd import synthetic
from adv_grpo_amd.config.experiments import get_config
from adv_grpo_amd.mmdit_train import SD3TransformerLoRA
from adv_grpo_amd.model_configs import ClipConfig, MMDiTConfig, VaeConfig
from adv_grpo_amd.pickscore_scorer import PickScoreScorer
from adv_grpo_amd.pipeline import SD3Pipeline
from adv_grpo_amd.trainer import SyntheticData, Trainer
from adv_grpo_amd.vae import AutoencoderKLDecoder
dev = torch.device("cuda", 0)
cfg = get_config("pickscore_cotrain_sd3_fast", gpu_number=1)
cfg.sample.num_image_per_prompt = 8; cfg.sample.num_batches_per_epoch = 2; cfg.train.gradient_accumulation_steps = 1; cfg.train_d = False
with synthetic.on_device(dev):
    tr = SD3TransformerLoRA(synthetic.mmdit_weights(MMDiTConfig(), 1234), MMDiTConfig(), dev, seed=cfg.seed)
    vae = AutoencoderKLDecoder(synthetic.vae_decoder_weights(VaeConfig(), 4321), VaeConfig(), dev)
    scorer = PickScoreScorer(dev, dtype=torch.bfloat16, model_sd=synthetic.clip_weights(ClipConfig(), 777), clip_cfg=ClipConfig())
trainer = Trainer(cfg, SD3Pipeline(tr, vae, dev), SyntheticData(resolution=cfg.resolution, device=dev), scorer, None, 0, 1, log_path=None)
trainer.run_epoch()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    trainer.run_epoch()
torch.cuda.set_sync_debug_mode("default")
c = collections.Counter((str(x.filename).split("/")[-1], x.lineno) for x in w if "synchroniz" in str(x.message).lower())
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(v, k)
print("timers", trainer.timers)
