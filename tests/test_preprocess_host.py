"""CPU: the host-built Pillow coefficient tables reproduce PIL's 8-bit antialiased bicubic resize exactly."""
import numpy as np
from PIL import Image


def _emulate(img, out):
    from adv_grpo_amd.preprocess import PRECISION_BITS, pil_bicubic_tables
    H, W = img.shape
    bh, ch, _ = pil_bicubic_tables(W, out)
    tmp = np.zeros((H, out), dtype=np.uint8)
    for ox in range(out):
        x0, n = bh[ox]
        ss = (1 << (PRECISION_BITS - 1)) + (img[:, x0:x0 + n].astype(np.int64) * ch[ox, :n]).sum(1)
        tmp[:, ox] = np.clip(ss >> PRECISION_BITS, 0, 255)
    bv, cv, _ = pil_bicubic_tables(H, out)
    res = np.zeros((out, out), dtype=np.uint8)
    for oy in range(out):
        y0, n = bv[oy]
        ss = (1 << (PRECISION_BITS - 1)) + (tmp[y0:y0 + n].astype(np.int64) * cv[oy, :n, None]).sum(0)
        res[oy] = np.clip(ss >> PRECISION_BITS, 0, 255)
    return res


def test_pil_bicubic_tables_bit_exact():
    rng = np.random.RandomState(0)
    for (h, w, out) in [(512, 512, 224), (256, 256, 224), (300, 300, 224), (224, 224, 224), (100, 100, 224)]:
        img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((out, out), Image.BICUBIC))
        assert np.array_equal(_emulate(img, out), ref), (h, w, out)
