"""QwenImageLayeredPipeline on the CDNA4 kernels — the DiT / VAE side of the reference's image-to-layers pipeline
(vllm_omni/diffusion/models/qwen_image/pipeline_qwen_image_layered.py:166-883): one input image is decomposed into `layers`
RGBA-style layers; the DiT denoises `layers + 1` frames (frame 0 = the recomposed image) next to the VAE-encoded input.

Relative to the Edit pipeline:
  * the noise latents are `[B, layers + 1, 16, h, w]`, packed frame by frame onto ONE sequence axis (:518-524), followed by the
    packed condition-image latents: img_shapes = [(1, h/2, w/2)] * (layers + 1) + [(1, h_c/2, w_c/2)] (:795-805);
  * the transformer is built with `use_layer3d_rope` (frame l at position l, the condition image at -1,
    qwen_image_transformer.py:65-176) and `use_additional_t_cond` (conditioning += addition_t_embedding[is_rgb], is_rgb = 0,
    :47-62; pipeline :834) — both flags come from `transformer/config.json` (:210-219);
  * the time shift uses mu = sqrt(S_cond / 256) and sigmas = linspace(1, 0, N + 1)[:-1] (:808-816);
  * the true-CFG combination is NOT norm-rescaled unless the request says `cfg_normalize` (:599-605, default False :662);
  * the decode drops frame 0 and decodes every layer frame on its own (:858-871): a request returns [layers, 3, H, W].
NOT built: `zero_cond_t` — the third Layered flag.  In the reference snapshot the block drops `modulate_index`
(qwen_image_transformer.py:552-564), so its behaviour is undefined; constructing the transformer with it raises and says so.
Request fields (the reference reads them off the request object, :667-679): `req.extra` keys `image` (the input picture, a
[1, C, H, W] / [1, C, 1, H, W] tensor in [-1, 1] already resized by `preprocess`, or pre-computed `image_latents` +
`image_latent_grid`), `layers` (default 4), `resolution` (640 | 1024), `cfg_normalize`, `prompt_image`.  An empty prompt asks
the reference for an auto-caption from its Qwen2.5-VL (`get_image_caption`, :495-516); here that needs a text encoder with a
`generate` capable model and is otherwise an error that says so."""
from __future__ import annotations

import numpy as np
import torch

from ...request import OmniDiffusionRequest
from .pipeline_qwen_image import BF16, get_qwen_image_post_process_func  # noqa: F401  (post-process: registry)
from .pipeline_qwen_image_edit import QwenImageEditPipeline, calculate_dimensions


def preprocess(image_size: tuple[int, int], resolution: int = 640, vae_scale_factor: int = 8) -> dict:
    """The pre-process arithmetic of the reference (:67-99) for an input of (width, height): the picture is resized to
    `calculated_*` (~resolution^2 at its aspect ratio, multiples of 32) for the VAE and the vision tower; the generated layers
    are `height x width` = the same rounded down to multiples of 16."""
    if resolution not in (640, 1024):
        raise ValueError(f"resolution must be either 640 or 1024, but got {resolution}")
    cw, ch, _ = calculate_dimensions(resolution * resolution, image_size[0] / image_size[1])
    m = vae_scale_factor * 2
    return {"calculated_width": cw, "calculated_height": ch, "width": cw // m * m, "height": ch // m * m}


def get_qwen_image_layered_pre_process_func(od_config=None):
    """The reference's request pre-processing for this pipeline (`get_qwen_image_layered_pre_process_func`, :42-105; run by the
    engine before the request reaches a worker): a PIL picture in `req.extra["image"]` is resized to ~resolution^2 at its own
    aspect ratio (`calculate_dimensions`, multiples of 32; diffusers `VaeImageProcessor.resize` -> LANCZOS), kept as
    `extra["prompt_image"]` (what the captioner sees), converted to a [1, 3, H, W] tensor in [-1, 1] (`VaeImageProcessor.preprocess`)
    and the generated size `req.height / req.width` is set to the same floored to multiples of 16.  Tensors pass through."""
    from .text_encoder import resize_picture

    def pre_process_func(requests):
        for req in requests:
            extra = req.extra if req.extra is not None else {}
            img = extra.get("image")
            if img is None or isinstance(img, torch.Tensor):
                continue
            if isinstance(img, (list, tuple)):
                img = img[0]
            plan = preprocess(img.size, int(extra.get("resolution") or 640))
            pic = resize_picture(img.convert("RGB"), plan["calculated_height"], plan["calculated_width"])
            arr = torch.from_numpy(np.asarray(pic, dtype=np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
            extra.update(image=arr * 2.0 - 1.0, prompt_image=extra.get("prompt_image", pic))
            req.extra = extra
            req.height, req.width = plan["height"], plan["width"]
        return requests

    return pre_process_func


class QwenImageLayeredPipeline(QwenImageEditPipeline):
    DEFAULT_LAYERS = 4

    def __init__(self, *, od_config=None, prefix: str = "", device=None, transformer=None, vae=None, **kw):
        tk = dict(kw.pop("transformer_kwargs", None) or {})
        tfc = getattr(od_config, "tf_model_config", None) if od_config is not None else None
        # transformer/config.json carries the flags (reference :210-219); a bare construction (tests, random weights) gets the
        # Layered defaults for the two that are built
        for flag in ("use_additional_t_cond", "use_layer3d_rope"):
            if flag not in tk and not (tfc is not None and flag in tfc):
                tk[flag] = True
        super().__init__(od_config=od_config, prefix=prefix, device=device, transformer=transformer, vae=vae,
                         transformer_kwargs=tk, **kw)
        tr = self.transformer
        if not (getattr(tr, "use_layer3d_rope", False) and getattr(tr, "use_additional_t_cond", False)):
            raise ValueError("the Layered pipeline needs a transformer built with use_layer3d_rope and use_additional_t_cond")

    # ------------------------------------------------------------------ helpers with the reference's semantics
    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width, layers=None):
        """[B, L, C, H, W] -> [B, L*(H/2)(W/2), 4C] (:518-524).  Called with 4-D input (the Edit base class packing ONE
        image's latents) it is the text-to-image packing."""
        if layers is None:
            return QwenImageEditPipeline._pack_latents(latents, batch_size, num_channels_latents, height, width)
        x = latents.view(batch_size, layers, num_channels_latents, height // 2, 2, width // 2, 2).permute(0, 1, 3, 5, 2, 4, 6)
        return x.reshape(batch_size, layers * (height // 2) * (width // 2), num_channels_latents * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, layers, vae_scale_factor=8):
        """[B, (layers + 1) * S, 4C] -> [B, C, layers + 1, H/8, W/8] (:526-541)."""
        B, _, ch = latents.shape
        h = 2 * (int(height) // (vae_scale_factor * 2))
        w = 2 * (int(width) // (vae_scale_factor * 2))
        x = latents.view(B, layers + 1, h // 2, w // 2, ch // 4, 2, 2).permute(0, 1, 4, 2, 5, 3, 6)
        return x.reshape(B, layers + 1, ch // 4, h, w).permute(0, 2, 1, 3, 4)

    def load_text_encoder(self, model_dir: str, device=None) -> None:
        """The Layered prompt is TEXT ONLY (the text-to-image template, first 34 tokens dropped: reference :233-238,333-365), so
        the encoder is the T2I one (`text_encoder/` + `tokenizer/`); `processor/` (Qwen2VLProcessor) is loaded as well when the
        checkpoint has it — only the auto-caption needs it."""
        import os

        from .pipeline_qwen_image import QwenImagePipeline

        QwenImagePipeline.load_text_encoder(self, model_dir, device)
        pr_dir = os.path.join(model_dir, "processor")
        if os.path.isdir(pr_dir):
            from transformers import AutoProcessor

            self._caption_processor = AutoProcessor.from_pretrained(pr_dir, local_files_only=True)

    _caption_processor = None

    def _encode_text(self, prompts: list[str]):
        return self.text_encoder.get_qwen_prompt_embeds(prompts, device=self.device)

    def get_image_caption(self, prompt_image, use_en_prompt: bool = True) -> str:
        """reference :495-516: the text encoder itself writes the prompt when the request has none."""
        enc, proc = self.text_encoder, self._caption_processor
        if enc is None or proc is None or not hasattr(getattr(enc, "model", None), "generate"):
            raise NotImplementedError("an empty prompt asks for an auto-caption (reference get_image_caption): that needs the "
                                      "checkpoint's Qwen2.5-VL text encoder AND its processor/ folder; pass a prompt or prompt_embeds")
        from .text_encoder import to_processor_image

        text = IMAGE_CAPTION_PROMPT_EN if use_en_prompt else IMAGE_CAPTION_PROMPT_CN
        dev = next(enc.model.parameters()).device
        inputs = proc(text=text, images=to_processor_image(prompt_image), padding=True, return_tensors="pt").to(dev)
        ids = enc.model.generate(**inputs, max_new_tokens=512)
        trimmed = [o[len(i):] for i, o in zip(inputs.input_ids, ids)]
        return proc.batch_decode(trimmed, skip_special_tokens=True, clean_up_tokenization_spaces=False)[0].strip()

    # ------------------------------------------------------------------ request level
    def _req_params(self, req: OmniDiffusionRequest):
        extra = req.extra or {}
        if req.height is None or req.width is None:
            img = extra.get("image")
            if isinstance(img, torch.Tensor):                 # generated size = the (already resized) input, floored to 16
                m = self.vae_scale_factor * 2
                req.height, req.width = int(img.shape[-2]) // m * m, int(img.shape[-1]) // m * m
        if (req.prompt is None or (isinstance(req.prompt, str) and req.prompt.strip() == "")) and req.prompt_embeds is None:
            pic = extra.get("prompt_image", extra.get("image"))
            if pic is None:
                raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
            req.prompt = self.get_image_caption(pic, bool(extra.get("use_en_prompt", False)))
        return super()._req_params(req)

    def resolve_request(self, req: OmniDiffusionRequest, index: int = 0) -> list[dict]:
        extra = req.extra or {}
        layers = int(extra.get("layers") or self.DEFAULT_LAYERS)
        if layers < 1:
            raise ValueError("layers must be >= 1")
        height, width, steps, cfg, do_cfg = self._req_params(req)
        # ---- condition image -> packed latents + its own grid (reference prepare_latents :457-481)
        if extra.get("image_latents") is not None:
            shape = extra.get("image_latent_grid")
            if shape is None:
                raise ValueError("image_latents need `image_latent_grid` = (h/16, w/16) of the condition image")
            cond, gh_c, gw_c = extra["image_latents"].reshape(-1, 64).to(self.device, BF16), int(shape[0]), int(shape[1])
        elif extra.get("image") is not None:
            z = self._encode_vae_image(extra["image"])
            _, Cz, _, hc, wc = z.shape
            cond, gh_c, gw_c = QwenImageEditPipeline._pack_latents(z[:, :, 0], 1, Cz, hc, wc)[0], hc // 2, wc // 2
        else:
            raise ValueError("the Layered pipeline needs req.extra['image'] (or pre-computed 'image_latents')")
        if cond.shape[0] != gh_c * gw_c:
            raise ValueError("condition-image latents do not match their token grid")
        # ---- the layers + 1 generated frames: noise drawn as [1, L + 1, 16, h, w] (:459-466,488-490)
        gh, gw = height // self.vae_scale_factor // 2, width // self.vae_scale_factor // 2
        S = (layers + 1) * gh * gw
        saved = (req.latents, req.height, req.width)
        noise = req.latents
        if noise is None:
            gen = req.generator
            if gen is None and req.seed is not None:
                gen = torch.Generator(device="cpu").manual_seed(req.seed)
            g0 = gen[0] if isinstance(gen, (list, tuple)) else gen
            n = int(req.num_outputs_per_prompt or 1) * (len(req.prompt) if isinstance(req.prompt, list) else 1)
            x = torch.randn((n, layers + 1, self.transformer.in_channels // 4, 2 * gh, 2 * gw), generator=g0,
                            device=g0.device if g0 is not None else self.device, dtype=BF16).to(self.device)
            noise = self._pack_latents(x, n, self.transformer.in_channels // 4, 2 * gh, 2 * gw, layers + 1)
        noise = noise.reshape(-1, S, noise.shape[-1])
        # the text-to-image resolver expands prompts / samples; the latents it sees are ours (S rows per sample)
        try:
            req.latents = noise
            req.height, req.width = height, width
            samples = self._resolve_with_rows(req, index, S)
        finally:
            req.latents, req.height, req.width = saved
        from .rope import layered_grids

        grid = layered_grids(tuple((1, gh, gw) for _ in range(layers + 1)) + ((1, gh_c, gw_c),))   # explicit frame indices
        for sm in samples:
            sm.update(cond=cond, grid=grid, layers=layers, t_cond=0,
                      cfg_normalize=bool(extra.get("cfg_normalize", False)),
                      mu=float((cond.shape[0] / (256 * 256 / 16 / 16)) ** 0.5),
                      sigmas=np.linspace(1.0, 0, steps + 1)[:-1])
        return samples

    def _resolve_with_rows(self, req: OmniDiffusionRequest, index: int, S: int) -> list[dict]:
        """QwenImagePipeline.resolve_request with `S` latent rows per sample instead of (h/16)(w/16)."""
        from .pipeline_qwen_image import QwenImagePipeline

        height, width, steps, cfg, do_cfg = QwenImagePipeline._req_params(self, req)
        n = int(req.num_outputs_per_prompt or 1)
        pos = self._rows_of(req.prompt_embeds, req.prompt_embeds_mask, req.prompt, n)
        neg = None
        if do_cfg:
            if req.negative_prompt_embeds is None and self.text_encoder is None:
                raise NotImplementedError("true-CFG needs negative_prompt_embeds (no text encoder is loaded)")
            neg_prompt = req.negative_prompt if req.negative_prompt is not None else ""
            if isinstance(neg_prompt, str) and req.negative_prompt_embeds is None:
                neg_prompt = [neg_prompt] * (len(pos) // n)
            neg = self._rows_of(req.negative_prompt_embeds, req.negative_prompt_embeds_mask, neg_prompt, n)
            if len(neg) != len(pos):
                raise ValueError(f"{len(pos) // n} prompts but {len(neg) // n} negative prompts")
        lat_all = req.latents.reshape(-1, S, req.latents.shape[-1])
        if lat_all.shape[0] not in (1, len(pos)):
            raise ValueError(f"latents carry {lat_all.shape[0]} samples, the request expands to {len(pos)}")
        return [dict(req=index, k=k, height=height, width=width, steps=steps, cfg=float(cfg), do_cfg=do_cfg, grid=None,
                     lat=lat_all[k if lat_all.shape[0] > 1 else 0].to(self.device, BF16), pos=pos[k],
                     neg=neg[k] if do_cfg else None) for k in range(len(pos))]

    # ------------------------------------------------------------------ data-parallel worker hooks (GPUWorker.execute_model)
    def _layers_of(self, req: OmniDiffusionRequest) -> int:
        return int((req.extra or {}).get("layers") or self.DEFAULT_LAYERS)

    def latent_rows(self, req: OmniDiffusionRequest) -> int:
        """Packed-latent rows of one finished sample: (layers + 1) frames of (h/16)(w/16) tokens."""
        height, width, *_ = self._req_params(req)
        return (self._layers_of(req) + 1) * (height // 16) * (width // 16)

    def images_per_sample(self, req: OmniDiffusionRequest) -> int:
        return self._layers_of(req)

    def decode_request(self, req: OmniDiffusionRequest, lat: torch.Tensor) -> torch.Tensor:
        """[n, (layers + 1) * S, 64] of `req` -> [n * layers, 3, H, W] (one image per layer, frame 0 dropped)."""
        height, width, *_ = self._req_params(req)
        return self._decode_samples(lat, dict(layers=self._layers_of(req), height=height, width=width))

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def _decode_samples(self, lat: torch.Tensor, sample: dict) -> torch.Tensor:
        """[n, (layers + 1) * S, 64] -> [n * layers, C, H, W]: frame 0 (the recomposed input) is dropped, every layer frame is
        decoded as an image of its own (:858-871), DECODE_BATCH of them per VAE call."""
        L = sample["layers"]
        z = self._unpack_latents(lat, sample["height"], sample["width"], L, self.vae_scale_factor).to(self.vae.dtype)
        mean = self._latents_mean.to(z.device, z.dtype)
        inv_std = 1.0 / self._latents_std.to(z.device, z.dtype)
        z = z / inv_std + mean
        b, c, f, h, w = z.shape
        z = z[:, :, 1:].permute(0, 2, 1, 3, 4).reshape(-1, c, 1, h, w)
        return torch.cat([self.vae.decode(z[i:i + self.DECODE_BATCH], return_dict=False)[0][:, :, 0]
                          for i in range(0, z.shape[0], self.DECODE_BATCH)])


IMAGE_CAPTION_PROMPT_EN = (  # reference :247-259
    "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"
    "<|im_start|>user\n# Image Annotator\nYou are a professional\nimage annotator. Please write an image caption based on the "
    "input image:\n1. Write the caption using natural,\ndescriptive language without structured formats or rich text.\n2. Enrich "
    "caption details by including: \n - Object\nattributes, such as quantity, color, shape, size, material, state, position, "
    "actions, and so on\n - Vision Relations\nbetween objects, such as spatial relations, functional relations, possessive "
    "relations, attachment relations, action\nrelations, comparative relations, causal relations, and so on\n - Environmental "
    "details, such as weather, lighting,\ncolors, textures, atmosphere, and so on\n - Identify the text clearly visible in the "
    "image, without translation or\nexplanation, and highlight it in the caption with quotation marks\n3. Maintain authenticity "
    "and accuracy:\n - Avoid\ngeneralizations\n - Describe all visible information in the image, while do not add information "
    "not explicitly shown in\nthe image\n<|vision_start|><|image_pad|><|vision_end|><|im_end|>\n<|im_start|>assistant\n")
IMAGE_CAPTION_PROMPT_CN = (  # reference :239-246
    "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n# 图像标注器\n你是一个专业的图像标注器。请基于输入图像，撰写图注:\n1.\n"
    "使用自然、描述性的语言撰写图注，不要使用结构化形式或富文本形式。\n2. 通过加入以下内容，丰富图注细节：\n - 对象的属性：如数量、颜色、形状、大小、位置、材质、状态、动作等\n -\n"
    "对象间的视觉关系：如空间关系、功能关系、动作关系、从属关系、比较关系、因果关系等\n - 环境细节：例如天气、光照、颜色、纹理、气氛等\n - 文字内容：识别图像中清晰可见的文字，不做翻译和解释，"
    "用引号在图注中强调\n3.\n保持真实性与准确性：\n - 不要使用笼统的描述\n -\n描述图像中所有可见的信息，但不要加入没有在图像中出现的内容\n"
    "<|vision_start|><|image_pad|><|vision_end|><|im_end|>\n<|im_start|>assistant\n")
