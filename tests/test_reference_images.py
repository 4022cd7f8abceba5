"""Reference-image store: file -> tensor arithmetic of TP:773-799, prefetch, fallback.  Host code."""
import json
import os

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")


def _write(tmp, name, size, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 256, (size[1], size[0], 3), dtype=np.uint8)).save(os.path.join(tmp, name))


def test_load_resize_to_tensor_prefetch_and_fallback(tmp_path):
    from PIL import Image
    from adv_grpo_amd.reference_images import ReferenceImageStore, load_image
    d = str(tmp_path)
    _write(d, "a0.png", (640, 480), 0); _write(d, "a1.png", (64, 64), 1); _write(d, "fb.png", (32, 32), 2)
    json.dump({"a cat": ["a0.png", "a1.png"], "a dog": ["missing.png"]}, open(os.path.join(d, "map.json"), "w"))
    # arithmetic: PIL bilinear (antialiased) resize to (R,R), uint8 -> /255, CHW
    t = load_image(os.path.join(d, "a0.png"), 64)
    ref = np.asarray(Image.open(os.path.join(d, "a0.png")).convert("RGB").resize((64, 64), Image.BILINEAR), dtype=np.float32) / 255
    assert t.shape == (3, 64, 64) and np.array_equal(t.permute(1, 2, 0).numpy(), ref)
    st = ReferenceImageStore(os.path.join(d, "map.json"), d, resolution=64, device="cpu", fallback_path=os.path.join(d, "fb.png"))
    st.prefetch(["a cat", "a dog", "unknown"])
    cat = st.get("a cat")
    assert cat.shape == (2, 3, 64, 64) and cat.dtype == torch.float32 and torch.equal(cat[0], t)
    assert torch.equal(st.get("a cat", n=1), cat[:1])                       # second call: no pending future, loads again
    dog = st.get("a dog")                                                   # unreadable file -> fallback image (TP:781-785)
    assert torch.equal(dog[0], load_image(os.path.join(d, "fb.png"), 64))
    with pytest.raises(KeyError):
        st.get("unknown")
    st2 = ReferenceImageStore(os.path.join(d, "map.json"), d, resolution=64, device="cpu")
    with pytest.raises(FileNotFoundError):
        st2.get("a dog")
    st.close(); st2.close()


def test_prompt_file_data_interface(tmp_path):
    from adv_grpo_amd.trainer import PromptFileData
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "ds")); os.makedirs(os.path.join(d, "img"))
    open(os.path.join(d, "ds", "train.txt"), "w").write("a cat\na dog\n")
    for i in range(2):
        _write(os.path.join(d, "img"), f"c{i}.png", (48, 48), i); _write(os.path.join(d, "img"), f"d{i}.png", (48, 48), 10 + i)
    json.dump({"a cat": ["c0.png", "c1.png"], "a dog": ["d0.png", "d1.png"]}, open(os.path.join(d, "map.json"), "w"))
    calls = []
    def embed(prompts):
        calls.append(tuple(prompts))
        return torch.full((len(prompts), 5, 8), float(len(prompts[0]))), torch.zeros(len(prompts), 4)
    data = PromptFileData(os.path.join(d, "ds"), os.path.join(d, "map.json"), os.path.join(d, "img"), embed,
                          lambda ps: torch.arange(77)[None].repeat(len(ps), 1), resolution=32, device="cpu")
    assert len(data) == 2 and calls == [("",)]                      # negative prompt embedded once (TP:669)
    data.prefetch([0, 1])
    pe, ppe = data.prompt(1)
    assert pe.shape == (1, 5, 8) and float(pe[0, 0, 0]) == 5.0 and data.prompt(1)[0] is pe and len(calls) == 2   # cached
    assert data.reference_images(1, 2).shape == (2, 3, 32, 32) and data.clip_ids(0, 3).shape == (3, 77)
