"""OCR reward (host plugin).  Mirror of ``OcrScorer`` (adv_grpo/ocr.py:8-65): the target text is what the prompt puts in
double quotes, the reward is ``1 - min(Levenshtein(recognised, target), len(target)) / len(target)`` after removing blanks
and lower-casing both, 0 distance when the target is a substring of what was read, maximum penalty when recognition fails.

Text recognition itself is not part of the accelerated path (host plugin, no kernel): the reference
calls PaddleOCR on the CPU.  PaddleOCR is used here too when it is importable; otherwise a ``recognizer`` callable
(``uint8 HWC ndarray -> str``) must be supplied -- nothing is silently skipped.
"""
import numpy as np


def levenshtein(a: str, b: str) -> int:
    """Edit distance (insert / delete / substitute, unit costs) -- what ``Levenshtein.distance`` returns (ocr.py:4,49)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class OcrScorer:
    def __init__(self, use_gpu=False, recognizer=None):
        self.recognizer = recognizer
        if recognizer is None:
            try:
                from paddleocr import PaddleOCR                                   # ocr.py:14-19
            except ImportError as e:
                raise RuntimeError("the ocr scorer needs paddleocr (not installed here) or an explicit recognizer: "
                                   "rewards.configure_ocr(lambda uint8_hwc_image: 'text')") from e
            ocr = PaddleOCR(use_angle_cls=False, lang="en", use_gpu=use_gpu, show_log=False)

            def paddle(img):
                result = ocr.ocr(img, cls=False)
                return "".join([res[1][0] if res[1][1] > 0 else "" for res in result[0]]) if result[0] else ""   # ocr.py:43
            self.recognizer = paddle

    def __call__(self, images, prompts):
        prompts = [prompt.split('"')[1] for prompt in prompts]                    # ocr.py:32
        assert len(images) == len(prompts), "Images and prompts must have the same length"
        rewards = []
        for img, prompt in zip(images, prompts):
            img = np.asarray(img)
            prompt = prompt.replace(" ", "").lower()
            try:
                text = self.recognizer(img).replace(" ", "").lower()
                dist = 0 if prompt in text else levenshtein(text, prompt)        # ocr.py:46-49
                if dist > len(prompt):                                           # ocr.py:52-53
                    dist = len(prompt)
            except Exception as e:                                                # ocr.py:55-58
                print(f"OCR processing failed: {str(e)}")
                dist = len(prompt)
            rewards.append(1 - dist / len(prompt))
        return rewards
