"""Architecture constants of the models on the hot path (public HF configs)."""
from dataclasses import dataclass


@dataclass
class MMDiTConfig:  # stabilityai/stable-diffusion-3.5-medium transformer ("MMDiT-X")
    num_layers: int = 24
    num_heads: int = 24
    head_dim: int = 64
    in_channels: int = 16
    out_channels: int = 16
    patch_size: int = 2
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 2048
    pos_embed_max_size: int = 384
    dual_attention_layers: tuple = tuple(range(13))
    qk_norm: bool = True

    @property
    def dim(self):
        return self.num_heads * self.head_dim


@dataclass
class VaeConfig:  # SD3 VAE
    latent_channels: int = 16
    block_out_channels: tuple = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 1.5305
    shift_factor: float = 0.0609


@dataclass
class ClipConfig:  # yuvalkirstain/PickScore_v1 = CLIP ViT-H/14
    v_hidden: int = 1280
    v_layers: int = 32
    v_heads: int = 16
    v_mlp: int = 5120
    image_size: int = 224
    patch: int = 14
    t_hidden: int = 1024
    t_layers: int = 24
    t_heads: int = 16
    t_mlp: int = 4096
    vocab: int = 49408
    max_pos: int = 77
    proj: int = 1024
    eos_token_id: int = 49407
    act: str = "gelu"


@dataclass
class ClipTextConfig:  # SD3's text_encoder (CLIP ViT-L/14 text tower, the defaults) / text_encoder_2 (OpenCLIP bigG: ClipTextConfig.bigg())
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    proj: int = 768
    vocab: int = 49408
    max_pos: int = 77
    act: str = "quick_gelu"
    eos_token_id: int = 2          # the released config.json says 2; pooling then takes argmax(input_ids) (text_encoders.py)

    @classmethod
    def bigg(cls):
        return cls(hidden=1280, layers=32, heads=20, mlp=5120, proj=1280, act="gelu")


@dataclass
class T5Config:  # SD3's text_encoder_3: google/t5-v1_1-xxl encoder
    d_model: int = 4096
    layers: int = 24
    heads: int = 64
    d_kv: int = 64
    d_ff: int = 10240
    vocab: int = 32128
    num_buckets: int = 32
    max_distance: int = 128


@dataclass
class DinoConfig:  # timm vit_base_patch14_dinov2.lvd142m
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    image_size: int = 518
    patch: int = 14


@dataclass
class QwenMMDiTConfig:  # Qwen/Qwen-Image transformer (diffusers QwenImageTransformer2DModel; BASELINE config 5)
    num_layers: int = 60
    num_heads: int = 24
    head_dim: int = 128
    in_channels: int = 64           # 2x2-packed latents of the 16-channel VAE
    out_channels: int = 16
    patch_size: int = 2
    joint_attention_dim: int = 3584
    axes_dims_rope: tuple = (16, 56, 56)
    rope_theta: float = 10000.0
    scale_rope: bool = True

    @property
    def dim(self):
        return self.num_heads * self.head_dim


@dataclass
class QwenVaeConfig:  # Qwen/Qwen-Image vae (diffusers AutoencoderKLQwenImage: Wan-2.1-style causal 3-D VAE; BASELINE config 5)
    base_dim: int = 96
    z_dim: int = 16
    dim_mult: tuple = (1, 2, 4, 4)
    num_res_blocks: int = 2
    latents_mean: tuple = (-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                           0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921)
    latents_std: tuple = (2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
                          3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160)

    @property
    def dims(self):
        m = self.dim_mult
        return [self.base_dim * u for u in (m[-1],) + tuple(reversed(m))]     # [384, 384, 384, 192, 96]

    def up_block_io(self, i):
        """(input width, output width, has upsampler) of decoder.up_blocks[i]: every upsampler halves the width."""
        d = self.dims
        return (d[i] if i == 0 else d[i] // 2), d[i + 1], i != len(self.dim_mult) - 1


@dataclass
class QwenTextConfig:  # Qwen/Qwen-Image text_encoder = Qwen2.5-VL-7B-Instruct's language model (transformers Qwen2_5_VLTextModel; BASELINE config 5)
    vocab_size: int = 152064
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_layers: int = 28
    num_heads: int = 28
    num_kv_heads: int = 4
    rms_eps: float = 1e-6
    rope_theta: float = 1e6

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads
