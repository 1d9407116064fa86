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
        hidden = out.hidden_states[-1]
        mask = toks.attention_mask.bool()
        lens = mask.sum(dim=1).tolist()
        split = [h[drop:] for h in torch.split(hidden[mask], lens, dim=0)]       # _extract_masked_hidden + drop (:351-357,380)
        T = max(e.shape[0] for e in split)
        emb = torch.stack([torch.cat([u, u.new_zeros(T - u.shape[0], u.shape[1])]) for u in split])
        msk = torch.stack([torch.cat([torch.ones(u.shape[0], dtype=torch.long, device=u.device),
                                      torch.zeros(T - u.shape[0], dtype=torch.long, device=u.device)]) for u in split])
        return emb.to(device=dev, dtype=dtype or self.dtype), msk.to(dev)
