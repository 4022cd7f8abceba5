import sys, os
mode = sys.argv[1]
sys.argv = ["x", "60"] + sys.argv[2:]
sys.path.insert(0, ".")
import torch
from adv_grpo_amd import ops
from adv_grpo_amd.qwen_mmdit_train import QwenImageTransformerLoRA
if mode == "nooverlap":
    orig = QwenImageTransformerLoRA.__init__
    def init(self, *a, **k):
        orig(self, *a, **k); self.overlap_wgrad = False
    QwenImageTransformerLoRA.__init__ = init
elif mode == "prio":
    oc = ops.concurrent_stream
    def cs(device, partners=()):
        return torch.cuda.Stream(device=device, priority=-1)
    ops.concurrent_stream = cs
    import adv_grpo_amd.qwen_mmdit_train as q
    q.ops.concurrent_stream = cs
print("mode", mode, flush=True)
exec(open("scripts/bench_gstep_qwen.py").read())
