"""TEST INFRASTRUCTURE (not product code): stand-ins for what a Qwen-Image-Edit checkpoint ships and this container cannot
download — the `processor/` folder (Qwen2VLProcessor = Qwen2 tokenizer + Qwen2VL image processor) and the 7B
Qwen2.5-VL weights — so that the reference's Edit prompt encoding (pipeline_qwen_image_edit.py:352-397,
pipeline_qwen_image_edit_plus.py:274-330) and the product's can be run on the same inputs.

  * `StubVLProcessor`: the call signature the pipelines use (`processor(text=, images=, padding=True, return_tensors="pt")`
    -> .input_ids / .attention_mask / .pixel_values / .image_grid_thw, `.to(device)`).  Pixels go through HF's REAL
    `Qwen2VLImageProcessorPil` (resize to a multiple of 28, rescale, normalise, 14-px patches, 2x2 merge); text goes through a
    deterministic word tokenizer in which the template's system block + "<|im_start|>user\n" is EXACTLY 64 ids — what the Qwen2
    tokenizer produces and what `prompt_template_encode_start_idx = 64` relies on — and every "<|image_pad|>" expands to
    t*h*w/4 image-token ids, as Qwen2VLProcessor does.
  * `make_random_vl_model`: a seeded, small HF `Qwen2_5_VLForConditionalGeneration` (vision tower + language model)."""
from __future__ import annotations

import re
import types

import torch

IMAGE_TOKEN, VISION_START, VISION_END, IM_START, IM_END, NEWLINE = 500, 501, 502, 503, 504, 505
VOCAB = 512
PREFIX_IDS = 64
_SPECIAL = {"<|vision_start|>": VISION_START, "<|vision_end|>": VISION_END, "<|im_start|>": IM_START, "<|im_end|>": IM_END}


class _Features:
    """What the processor returns: attribute access + `.to(device)` (BatchFeature's surface as the pipelines use it)."""

    def __init__(self, feats: dict):
        self.__dict__.update(feats)

    def to(self, device):
        return _Features({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.__dict__.items()})


class StubVLProcessor:
    def __init__(self, min_pixels: int = 56 * 56, max_pixels: int = 28 * 28 * 16):
        from transformers import Qwen2VLImageProcessorPil

        self.image_processor = Qwen2VLImageProcessorPil(patch_size=14, merge_size=2, temporal_patch_size=2,
                                                        min_pixels=min_pixels, max_pixels=max_pixels)
        self.calls = []

    def _ids(self, text: str, n_img_tokens: list[int]) -> list[int]:
        marker = "<|im_start|>user\n"
        cut = text.index(marker) + len(marker)
        ids = [400 + i for i in range(PREFIX_IDS)]            # the system block + "<|im_start|>user\n": 64 ids under Qwen2
        rest, k = text[cut:], 0
        for piece in re.split(r"(<\|vision_start\|>|<\|vision_end\|>|<\|image_pad\|>|<\|im_start\|>|<\|im_end\|>|\n)", rest):
            if not piece:
                continue
            if piece == "<|image_pad|>":
                ids += [IMAGE_TOKEN] * n_img_tokens[k]
                k += 1
            elif piece in _SPECIAL:
                ids.append(_SPECIAL[piece])
            elif piece == "\n":
                ids.append(NEWLINE)
            else:
                ids += [3 + (sum(w.encode()) * 7 + len(w)) % 380 for w in piece.split()]
        assert k == len(n_img_tokens), "number of <|image_pad|> markers != number of images"
        return ids

    def __call__(self, text, images=None, padding=True, return_tensors="pt"):
        text = [text] if isinstance(text, str) else list(text)
        self.calls.append(dict(text=text, n_images=0 if images is None else (len(images) if isinstance(images, list) else 1)))
        feats = {}
        n_tok: list[int] = []
        if images is not None:
            imgs = images if isinstance(images, list) else [images]
            px = self.image_processor(images=imgs, return_tensors="pt")
            feats["pixel_values"], feats["image_grid_thw"] = px["pixel_values"], px["image_grid_thw"]
            n_tok = [int(t * h * w) // 4 for t, h, w in px["image_grid_thw"].tolist()]
        if len(text) != 1 and n_tok:
            raise NotImplementedError("the stub serves one prompt per call when images are given (the pipelines' B=1 requests)")
        ids = [self._ids(t, n_tok) for t in text]
        L = max(len(x) for x in ids)
        feats["input_ids"] = torch.tensor([x + [0] * (L - len(x)) for x in ids], dtype=torch.long)
        feats["attention_mask"] = torch.tensor([[1] * len(x) + [0] * (L - len(x)) for x in ids], dtype=torch.long)
        return _Features(feats)


def make_random_vl_model(seed: int = 5, hidden: int = 64, dtype=torch.float32):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration

    half = hidden // 4 // 2                                     # rotary pairs per head (4 heads): the three M-RoPE sections sum to it
    sec = [half // 4, (half - half // 4) // 2]
    sec.append(half - sum(sec))
    cfg = Qwen2_5_VLConfig(
        text_config=dict(vocab_size=VOCAB, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                         intermediate_size=128, max_position_embeddings=4096, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                         rope_scaling={"type": "default", "mrope_section": sec, "rope_type": "default"}),
        vision_config=dict(depth=2, hidden_size=32, out_hidden_size=hidden, num_heads=2, intermediate_size=64, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, window_size=56, fullatt_block_indexes=[1], in_channels=3),
        image_token_id=IMAGE_TOKEN, video_token_id=499, vision_start_token_id=VISION_START, vision_end_token_id=VISION_END)
    torch.manual_seed(seed)
    return Qwen2_5_VLForConditionalGeneration(cfg).to(dtype).eval(), cfg


def test_images(n: int = 2):
    """Deterministic RGB test pictures (PIL), different sizes / aspect ratios."""
    import numpy as np
    from PIL import Image

    rng = np.random.RandomState(3)
    sizes = [(96, 64), (70, 112), (128, 128)]
    return [Image.fromarray(rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)) for (w, h) in sizes[:n]]
