"""QwenImageEditPipeline on the CDNA4 kernels — the DiT / VAE side of the reference's image-editing pipeline
(vllm_omni/diffusion/models/qwen_image/pipeline_qwen_image_edit.py:194-830).

Same transformer as text-to-image; what changes:
  * the condition image is VAE-ENCODED (posterior mean, `sample_mode="argmax"`, :459-467), normalised with the VAE's
    latents_mean / latents_std (:468-478) and packed like the noise latents (:519-522);
  * every DiT forward sees `torch.cat([latents, image_latents], dim=1)` (:600-603) with TWO entries in `img_shapes`
    (:770-777), so the condition image gets its own RoPE frame index, and the prediction is sliced back to the first
    `latents.size(1)` tokens (:632);
  * everything else — true-CFG, Flow-Match Euler, decode — is the text-to-image loop.
Prompt encoding for Edit runs the Qwen2.5-VL VISION tower on the image as well (:306-397): `text_encoder` is a
`QwenEditPromptEncoder` (text_encoder.py: HF Qwen2_5_VLForConditionalGeneration + the checkpoint's Qwen2VLProcessor); a request
with `prompt` and no `prompt_embeds` is encoded from the prompt AND the picture — `req.extra["prompt_image"]` (PIL / array /
tensor; what the reference calls `prompt_image`, the resized input, :686-700) or, by default, the condition image itself.
Requests may still carry `prompt_embeds`.  The image is passed as `req.extra["image"]`
([1, 3, H, W] or [1, 3, 1, H, W] in [-1, 1], already resized: the reference's pre-process picks a ~1024^2 area with the
image's aspect ratio, :59-98,124-132) or as pre-computed packed `req.extra["image_latents"]` [S_c, 64]."""
from __future__ import annotations

import math

import torch

from ...request import OmniDiffusionRequest
from .autoencoder_kl_qwenimage import AutoencoderKLQwenImage
from .pipeline_qwen_image import BF16, QwenImagePipeline


def calculate_dimensions(target_area: float, ratio: float) -> tuple[int, int, None]:
    """(width, height) of ~target_area pixels at aspect `ratio` = w / h, both multiples of 32 (reference :124-132)."""
    width = math.sqrt(target_area * ratio)
    height = width / ratio
    return round(width / 32) * 32, round(height / 32) * 32, None


class QwenImageEditPipeline(QwenImagePipeline):
    def __init__(self, *, od_config=None, prefix: str = "", device=None, transformer=None, vae=None, **kw):
        dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        super().__init__(od_config=od_config, prefix=prefix, device=dev, transformer=transformer,
                         vae=vae if vae is not None else AutoencoderKLQwenImage(device=dev, with_encoder=True), **kw)
        if not getattr(self.vae, "with_encoder", False):
            raise ValueError("the Edit pipeline needs a VAE built with its encoder (with_encoder=True)")

    @torch.no_grad()
    def _encode_vae_image(self, image: torch.Tensor) -> torch.Tensor:
        """[B, 3, (1,) H, W] in [-1, 1] -> normalised latents [B, 16, 1, H/8, W/8]   (reference :459-480)."""
        if image.dim() == 4:
            image = image.unsqueeze(2)
        z = self.vae.encode(image.to(self.device, BF16)).float()
        mean = self._latents_mean.to(z.device)
        std = self._latents_std.to(z.device)
        return ((z - mean) / std).to(BF16)

    def load_text_encoder(self, model_dir: str, device=None) -> None:
        """`text_encoder/` (Qwen2_5_VLForConditionalGeneration) + `processor/` (Qwen2VLProcessor) of an Edit checkpoint
        (reference :220-232), local files only."""
        import os

        from transformers import AutoProcessor, Qwen2_5_VLForConditionalGeneration

        from .text_encoder import QwenEditPromptEncoder

        te_dir, pr_dir = os.path.join(model_dir, "text_encoder"), os.path.join(model_dir, "processor")
        if not (os.path.isdir(te_dir) and os.path.isdir(pr_dir)):
            raise FileNotFoundError(f"{model_dir!r} has no text_encoder/ + processor/ folders")
        dev = torch.device(device if device is not None else self.device)
        model = Qwen2_5_VLForConditionalGeneration.from_pretrained(te_dir, torch_dtype=BF16, local_files_only=True).to(dev).eval()
        self.text_encoder = QwenEditPromptEncoder(model, AutoProcessor.from_pretrained(pr_dir, local_files_only=True), dtype=BF16,
                                                  multi_image=self._multi_image_prompt)

    _multi_image_prompt = False          # Edit-Plus numbers its pictures in the prompt ("Picture 1: ...")
    _prompt_images = None                # set for the duration of one resolve_request

    def _encode_text(self, prompts: list[str]):
        enc = self.text_encoder
        if not hasattr(enc, "processor"):                    # a text-only encoder was installed: usable, but blind to the picture
            return enc.get_qwen_prompt_embeds(prompts, device=self.device)
        if self._prompt_images is None:
            raise ValueError("encoding an Edit prompt needs the picture: req.extra['prompt_image'] or req.extra['image']")
        return enc.get_qwen_prompt_embeds(prompts, image=self._prompt_images, device=self.device)

    def _prompt_pictures(self, req: OmniDiffusionRequest):
        extra = req.extra or {}
        pics = extra.get("prompt_image", extra.get("image"))
        if isinstance(pics, (list, tuple)) and not self._multi_image_prompt:
            pics = pics[0]
        return pics

    def resolve_request(self, req: OmniDiffusionRequest, index: int = 0) -> list[dict]:
        self._prompt_images = self._prompt_pictures(req)
        try:
            samples = super().resolve_request(req, index)
        finally:
            self._prompt_images = None
        extra = req.extra or {}
        if extra.get("image_latents") is not None:
            packed, hc, wc = extra["image_latents"].reshape(-1, 64).to(self.device, BF16), None, None
            shape = extra.get("image_latent_grid")
            if shape is None:
                raise ValueError("image_latents need `image_latent_grid` = (h/16, w/16) of the condition image")
            gh_c, gw_c = int(shape[0]), int(shape[1])
        elif extra.get("image") is not None:
            z = self._encode_vae_image(extra["image"])                       # [1, 16, 1, h, w]
            _, Cz, _, hc, wc = z.shape
            packed = self._pack_latents(z[:, :, 0], 1, Cz, hc, wc)[0]           # [S_c, 64]
            gh_c, gw_c = hc // 2, wc // 2
        else:
            raise ValueError("the Edit pipeline needs req.extra['image'] (or pre-computed 'image_latents')")
        if packed.shape[0] != gh_c * gw_c:
            raise ValueError("condition-image latents do not match their token grid")
        for sm in samples:
            sm["cond"] = packed
            sm["grid"] = (sm["grid"], (1, gh_c, gw_c))                       # img_shapes = [[(1, h, w), (1, h_c, w_c)]]
        return samples
