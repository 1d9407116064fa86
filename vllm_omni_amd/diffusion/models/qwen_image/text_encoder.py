"""Prompt encoding for Qwen-Image — mirror of `_get_qwen_prompt_embeds` (reference
vllm_omni/diffusion/models/qwen_image/pipeline_qwen_image.py:351-392): the prompt is wrapped in the chat template, run
through the Qwen2.5-VL language model (HF `transformers`, as in the reference: `Qwen2_5_VLForConditionalGeneration`,
`output_hidden_states=True`, last hidden state), the 34 template tokens are dropped, sequences are zero-padded to the longest
and returned with their mask.

This is the REQUEST-SIDE BOUNDARY, not the DiT hot path: it runs once per request, on the same HF eager
kernels the reference uses; its output feeds `prompt_embeds` of the native denoise loop.  No checkpoint can be downloaded
here, so tests build a small random-weight Qwen2.5-VL text model and a byte-level tokenizer stand-in with the same call
signature (`ByteTokenizer`); with a real checkpoint pass `Qwen2Tokenizer` and the loaded model."""
from __future__ import annotations

import torch

PROMPT_TEMPLATE_ENCODE = ("<|im_start|>system\nDescribe the image by detailing the color, shape, size, texture, quantity, text, "
                          "spatial relationships of the objects and background:<|im_end|>\n<|im_start|>user\n{}<|im_end|>\n"
                          "<|im_start|>assistant\n")
PROMPT_TEMPLATE_ENCODE_START_IDX = 34            # tokens of the template prefix under the Qwen2 tokenizer (:282)


class ByteTokenizer:
    """Stand-in with the `Qwen2Tokenizer.__call__` signature used by the pipeline: UTF-8 bytes + 3 (0 = pad)."""

    pad_token_id = 0
    vocab_size = 259

    def __call__(self, text, max_length=None, padding=True, truncation=True, return_tensors="pt"):
        text = [text] if isinstance(text, str) else list(text)
        ids = [[b + 3 for b in t.encode("utf-8")] for t in text]
        if truncation and max_length:
            ids = [x[:max_length] for x in ids]
        L = max(len(x) for x in ids)
        input_ids = torch.tensor([x + [0] * (L - len(x)) for x in ids], dtype=torch.long)
        mask = torch.tensor([[1] * len(x) + [0] * (L - len(x)) for x in ids], dtype=torch.long)
        return _Tokens(input_ids, mask)

    def template_prefix_tokens(self, template: str) -> int:
        return len(template.split("{}")[0].encode("utf-8"))


class _Tokens:
    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def to(self, device):
        return _Tokens(self.input_ids.to(device), self.attention_mask.to(device))


class QwenPromptEncoder:
    def __init__(self, text_model, tokenizer, dtype=torch.bfloat16, tokenizer_max_length: int = 1024,
                 template: str = PROMPT_TEMPLATE_ENCODE, drop_idx: int | None = None):
        self.model, self.tokenizer, self.dtype = text_model, tokenizer, dtype
        self.tokenizer_max_length = tokenizer_max_length
        self.prompt_template_encode = template
        if drop_idx is None:
            drop_idx = tokenizer.template_prefix_tokens(template) if hasattr(tokenizer, "template_prefix_tokens") \
                else PROMPT_TEMPLATE_ENCODE_START_IDX
        self.prompt_template_encode_start_idx = drop_idx

    @classmethod
    def random_init(cls, hidden_size: int = 3584, num_layers: int = 2, num_heads: int = 28, num_kv_heads: int = 4,
                    intermediate_size: int = 512, device="cpu", dtype=torch.bfloat16, seed: int = 0):
        """Random-weight Qwen2.5-VL TEXT model of the real width (3584 -> joint_attention_dim) with few layers: what the
        end-to-end tests use in place of the 7B checkpoint."""
        from transformers import Qwen2_5_VLTextConfig, Qwen2_5_VLTextModel

        torch.manual_seed(seed)
        tok = ByteTokenizer()
        half = hidden_size // num_heads // 2                       # rotary pairs per head; Qwen2.5-VL: 64 -> [16, 24, 24]
        sec = [half // 4, (half - half // 4) // 2]
        sec.append(half - sum(sec))
        cfg = Qwen2_5_VLTextConfig(vocab_size=tok.vocab_size, hidden_size=hidden_size, num_hidden_layers=num_layers,
                                   num_attention_heads=num_heads, num_key_value_heads=num_kv_heads,
                                   intermediate_size=intermediate_size, max_position_embeddings=4096,
                                   bos_token_id=1, eos_token_id=2, pad_token_id=0,
                                   rope_scaling={"type": "default", "mrope_section": sec, "rope_type": "default"})
        model = Qwen2_5_VLTextModel(cfg).to(device=device, dtype=dtype).eval()
        return cls(model, tok, dtype=dtype)

    @torch.no_grad()
    def get_qwen_prompt_embeds(self, prompt, device=None, dtype=None):
        """-> (prompt_embeds [B, Tmax, hidden] zero-padded, mask [B, Tmax] long)   (reference :359-392)."""
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        dev = device if device is not None else next(self.model.parameters()).device
        drop = self.prompt_template_encode_start_idx
        txt = [self.prompt_template_encode.format(e) for e in prompt]
        toks = self.tokenizer(txt, max_length=self.tokenizer_max_length + drop, padding=True, truncation=True,
                              return_tensors="pt").to(next(self.model.parameters()).device)
        out = self.model(input_ids=toks.input_ids, attention_mask=toks.attention_mask, output_hidden_states=True)
        emb, msk = masked_drop_pad(out.hidden_states[-1], toks.attention_mask, drop)
        return emb.to(device=dev, dtype=dtype or self.dtype), msk.to(dev)


def masked_drop_pad(hidden: torch.Tensor, attention_mask: torch.Tensor, drop: int):
    """`_extract_masked_hidden` + drop of the template tokens + zero-padding to the longest sequence + mask, shared by the
    text-to-image (pipeline_qwen_image.py:351-357,380-390) and the Edit pipelines (pipeline_qwen_image_edit.py:298-305,
    383-393)."""
    mask = attention_mask.bool()
    lens = mask.sum(dim=1).tolist()
    split = [h[drop:] for h in torch.split(hidden[mask], lens, dim=0)]
    T = max(e.shape[0] for e in split)
    emb = torch.stack([torch.cat([u, u.new_zeros(T - u.shape[0], u.shape[1])]) for u in split])
    msk = torch.stack([torch.cat([torch.ones(u.shape[0], dtype=torch.long, device=u.device),
                                  torch.zeros(T - u.shape[0], dtype=torch.long, device=u.device)]) for u in split])
    return emb, msk


# ---- image-editing prompts: the Qwen2.5-VL VISION tower sees the picture(s) ------------------------------
EDIT_PROMPT_TEMPLATE_ENCODE = (     # reference pipeline_qwen_image_edit.py:240 (one picture, marker inside the template)
    "<|im_start|>system\nDescribe the key features of the input image (color, shape, size, texture, objects, background), then "
    "explain how the user's text instruction should alter or modify the image. Generate a new image that meets the user's "
    "requirements while maintaining consistency with the original input where appropriate.<|im_end|>\n<|im_start|>user\n"
    "<|vision_start|><|image_pad|><|vision_end|>{}<|im_end|>\n<|im_start|>assistant\n")
EDIT_PLUS_PROMPT_TEMPLATE_ENCODE = (  # reference pipeline_qwen_image_edit_plus.py:203-209 (markers come with the pictures)
    "<|im_start|>system\nDescribe the key features of the input image (color, shape, size, texture, objects, background), then "
    "explain how the user's text instruction should alter or modify the image. Generate a new image that meets the user's "
    "requirements while maintaining consistency with the original input where appropriate.<|im_end|>\n<|im_start|>user\n{}"
    "<|im_end|>\n<|im_start|>assistant\n")
EDIT_PROMPT_TEMPLATE_ENCODE_START_IDX = 64       # both Edit pipelines (:241 / :210)
EDIT_PLUS_IMG_PROMPT = "Picture {}: <|vision_start|><|image_pad|><|vision_end|>"     # edit_plus :285


def to_processor_image(x):
    """What the HF image processor takes: PIL images pass through; a tensor [3, H, W] / [1, 3, H, W] / [3, 1, H, W] /
    [1, 3, 1, H, W] in [-1, 1] (the layouts the VAE encoder takes: one picture, optional batch and frame axes of size 1) becomes
    a uint8 HWC array.  Anything else (a batch of several pictures, a wrong channel count) raises — one bad
    `req.extra['image']` must not hang or crash the worker."""
    if isinstance(x, torch.Tensor):
        t = x.detach().float().cpu()
        shape = tuple(t.shape)
        if t.dim() == 5 and t.shape[0] == 1 and t.shape[2] == 1:        # [1, 3, 1, H, W]
            t = t[0, :, 0]
        elif t.dim() == 4 and t.shape[0] == 1:                          # [1, 3, H, W]
            t = t[0]
        elif t.dim() == 4 and t.shape[1] == 1:                          # [3, 1, H, W]
            t = t[:, 0]
        if t.dim() != 3 or t.shape[0] != 3:
            raise ValueError(f"cannot interpret a tensor of shape {shape} as ONE RGB picture "
                             "([3,H,W], [1,3,H,W], [3,1,H,W] or [1,3,1,H,W])")
        return ((t.clamp(-1, 1) + 1) * 127.5).round().to(torch.uint8).permute(1, 2, 0).numpy()
    return x


def resize_picture(x, height: int, width: int):
    """diffusers `VaeImageProcessor.resize(image, height, width)` as the reference calls it for the vision-tower copies of the
    condition images (pipeline_qwen_image_edit_plus.py:116,650; third-party, restated: PIL -> `Image.resize((w, h), LANCZOS)`,
    tensor -> `F.interpolate(size=(h, w))`, i.e. nearest)."""
    if isinstance(x, torch.Tensor):
        t = x
        lead = t.dim()
        if lead == 3:
            t = t.unsqueeze(0)
        elif lead == 5:
            t = t[:, :, 0]
        t = torch.nn.functional.interpolate(t.float(), size=(height, width)).to(x.dtype)
        return t[0] if lead == 3 else (t.unsqueeze(2) if lead == 5 else t)
    from PIL import Image

    return x.resize((width, height), resample=Image.LANCZOS)


class QwenEditPromptEncoder:
    """`_get_qwen_prompt_embeds` of the Edit pipelines (pipeline_qwen_image_edit.py:352-397, multi_image=False;
    pipeline_qwen_image_edit_plus.py:274-330, multi_image=True): template -> `processor(text=, images=)` -> HF
    `Qwen2_5_VLForConditionalGeneration` with `pixel_values` / `image_grid_thw` (vision tower + language model),
    last hidden state, first 64 tokens dropped, zero-padded + mask.  `processor` is the checkpoint's `Qwen2VLProcessor`
    (tests: oracle/vl_stubs.StubVLProcessor, same call signature)."""

    def __init__(self, vl_model, processor, dtype=torch.bfloat16, multi_image: bool = False, template: str | None = None,
                 drop_idx: int = EDIT_PROMPT_TEMPLATE_ENCODE_START_IDX):
        self.model, self.processor, self.dtype, self.multi_image = vl_model, processor, dtype, multi_image
        self.prompt_template_encode = template or (EDIT_PLUS_PROMPT_TEMPLATE_ENCODE if multi_image else EDIT_PROMPT_TEMPLATE_ENCODE)
        self.prompt_template_encode_start_idx = drop_idx
        self.tokenizer_max_length = 1024

    @torch.no_grad()
    def get_qwen_prompt_embeds(self, prompt, image=None, device=None, dtype=None):
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        mdev = next(self.model.parameters()).device
        dev = device if device is not None else mdev
        if isinstance(image, (list, tuple)):
            image = [to_processor_image(im) for im in image]
        elif image is not None:
            image = to_processor_image(image)
        if self.multi_image:
            n = len(image) if isinstance(image, list) else (0 if image is None else 1)
            base = "".join(EDIT_PLUS_IMG_PROMPT.format(i + 1) for i in range(n))
            txt = [self.prompt_template_encode.format(base + e) for e in prompt]
        else:
            txt = [self.prompt_template_encode.format(e) for e in prompt]
        inputs = self.processor(text=txt, images=image, padding=True, return_tensors="pt").to(mdev)
        kw = {}
        if getattr(inputs, "pixel_values", None) is not None:
            kw = dict(pixel_values=inputs.pixel_values, image_grid_thw=inputs.image_grid_thw)
        out = self.model(input_ids=inputs.input_ids, attention_mask=inputs.attention_mask, output_hidden_states=True, **kw)
        emb, msk = masked_drop_pad(out.hidden_states[-1], inputs.attention_mask, self.prompt_template_encode_start_idx)
        return emb.to(device=dev, dtype=dtype or self.dtype), msk.to(dev)
