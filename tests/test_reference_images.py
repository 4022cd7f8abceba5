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


def test_eval_writer_round_trips_through_the_store(tmp_path):
    """scripts/eval.py:258-294: the eval loop's PNGs + prompt2img.json are exactly what the trainers read back as
    `json_path` / `reference_image_path` (TP:705-707,773-799): uint8 truncation, 512x512, {prompt: [file]}."""
    from PIL import Image
    from adv_grpo_amd.reference_images import ReferenceImageStore
    from adv_grpo_amd.trainer import write_eval_images, write_prompt2img
    g = torch.Generator().manual_seed(3)
    images = torch.rand(3, 3, 512, 512, generator=g)
    names = write_eval_images(images, str(tmp_path), rank=1, batch_idx=7)
    assert names == ["node0_rank1_00007_0.png", "node0_rank1_00007_1.png", "node0_rank1_00007_2.png"]
    prompts = ["a cat", "ein Hund ü", "一只鸟"]
    merged = write_prompt2img({p: [n] for p, n in zip(prompts, names)}, str(tmp_path))
    with open(tmp_path / "prompt2img.json", encoding="utf-8") as f:
        text = f.read()
    assert json.loads(text) == merged == {p: [n] for p, n in zip(prompts, names)}
    assert "一只鸟" in text and text.startswith("{\n  ")                 # ensure_ascii=False, indent=2
    want = (images.permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)                 # truncation, not rounding
    for n, w in zip(names, want):
        assert np.array_equal(np.asarray(Image.open(tmp_path / n)), w)                 # 512 -> 512 resize is the identity
    store = ReferenceImageStore(str(tmp_path / "prompt2img.json"), str(tmp_path), device="cpu")
    got = store.get(prompts[1], 1)
    assert torch.equal(got[0], torch.from_numpy(want[1]).permute(2, 0, 1).float() / 255)
    # a 256 px eval image is upsampled to 512 with PIL's default filter
    small = torch.rand(1, 3, 256, 256, generator=g)
    (name,) = write_eval_images(small, str(tmp_path), rank=0, batch_idx=0)
    ref = Image.fromarray((small[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).resize((512, 512))
    assert np.array_equal(np.asarray(Image.open(tmp_path / name)), np.asarray(ref))


def _merge_worker(rank, world, port, folder):
    import torch.distributed as dist
    from adv_grpo_amd.trainer import write_prompt2img
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    local = {f"prompt {rank}": [f"node0_rank{rank}_00000_0.png"], "shared": [f"node0_rank{rank}_00001_0.png"]}
    write_prompt2img(local, folder, world, rank)
    dist.barrier()
    dist.destroy_process_group()


def test_prompt2img_merges_ranks_in_rank_order(tmp_path):
    """gather_dict (scripts/eval.py:153-165): lists of a prompt seen on several ranks concatenate in rank order; only
    rank 0 writes."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_merge_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    with open(tmp_path / "prompt2img.json", encoding="utf-8") as f:
        got = json.load(f)
    assert got == {"prompt 0": ["node0_rank0_00000_0.png"], "shared": ["node0_rank0_00001_0.png", "node0_rank1_00001_0.png"],
                   "prompt 1": ["node0_rank1_00000_0.png"]}
    assert list(got) == ["prompt 0", "shared", "prompt 1"]
