"""QwenImageEditPlusPipeline on the CDNA4 kernels — the DiT / VAE side of the reference's multi-image editing pipeline
(vllm_omni/diffusion/models/qwen_image/pipeline_qwen_image_edit_plus.py:154-795).

Relative to QwenImageEditPipeline:
  * a request carries a LIST of condition images; each is resized on its own to ~1024^2 (`VAE_IMAGE_SIZE`) at its own
    aspect ratio for the VAE and to ~384^2 (`CONDITION_IMAGE_SIZE`) for the vision tower (:44-45,96-123);
  * every image is VAE-encoded and packed separately and the packed latents are concatenated on the sequence axis
    (:440-464); `img_shapes` gets one entry per image behind the generated image's (:729-738), so each condition image
    has its own RoPE frame index 1, 2, ...;
  * the prompt template numbers the images: "Picture 1: <|vision_start|><|image_pad|><|vision_end|>Picture 2: ..." in front
    of the user text (:286-299).
The denoise loop, true-CFG, slicing the prediction back to the generated image's tokens and the decode are the Edit /
text-to-image path.  Prompts are encoded through the Qwen2.5-VL vision tower with ALL pictures (QwenEditPromptEncoder,
multi_image=True; `req.extra["prompt_image"]` = the list at the vision-tower size, default: the condition images resized to
~384^2 as in the reference's pre-process); requests
may still carry `prompt_embeds`."""
from __future__ import annotations

import torch

from ...request import OmniDiffusionRequest
from .pipeline_qwen_image import BF16
from .pipeline_qwen_image_edit import QwenImageEditPipeline, calculate_dimensions

CONDITION_IMAGE_SIZE = 384 * 384        # reference :44
VAE_IMAGE_SIZE = 1024 * 1024            # reference :45
_IMG_PROMPT = "Picture {}: <|vision_start|><|image_pad|><|vision_end|>"
PROMPT_TEMPLATE_ENCODE = (              # reference :203-209
    "<|im_start|>system\nDescribe the key features of the input image "
    "(color, shape, size, texture, objects, background), then explain how the user's "
    "text instruction should alter or modify the image. Generate a new image that meets "
    "the user's requirements while maintaining consistency with the original input where "
    "appropriate.<|im_end|>\n<|im_start|>user\n{}<|im_end|>\n<|im_start|>assistant\n"
)


def edit_plus_prompt(prompt: str, n_images: int) -> str:
    """The text the reference hands to the Qwen2.5-VL processor for `n_images` condition images (:286-299)."""
    return PROMPT_TEMPLATE_ENCODE.format("".join(_IMG_PROMPT.format(i + 1) for i in range(n_images)) + prompt)


def plan_image_sizes(sizes: list[tuple[int, int]]) -> dict:
    """Pre-process arithmetic of the reference (:72-123) for images of (width, height) `sizes`: the generated image takes
    the FIRST image's aspect ratio at ~1024^2; every image gets a vision-tower size and a VAE size of its own."""
    w0, h0 = sizes[0]
    width, height, _ = calculate_dimensions(VAE_IMAGE_SIZE, w0 / h0)
    cond = [calculate_dimensions(CONDITION_IMAGE_SIZE, w / h)[:2] for w, h in sizes]
    vae = [calculate_dimensions(VAE_IMAGE_SIZE, w / h)[:2] for w, h in sizes]
    return {"width": width, "height": height, "condition_image_sizes": cond, "vae_image_sizes": vae}


def _picture_size(x) -> tuple[int, int]:
    """(width, height) of a PIL image or a [..., H, W] tensor."""
    if isinstance(x, torch.Tensor):
        return int(x.shape[-1]), int(x.shape[-2])
    return x.size


class QwenImageEditPlusPipeline(QwenImageEditPipeline):
    _multi_image_prompt = True

    def _prompt_pictures(self, req: OmniDiffusionRequest):
        """The pictures the vision tower sees.  `extra['prompt_image']` is used as given; by default the condition images are
        resized to ~384^2 at their own aspect ratio (`CONDITION_IMAGE_SIZE`), as the reference's pre-process does before prompt
        encoding (:96-123,637-650,699) — round 3 fed the VAE-sized images, so the embeddings differed from the reference's."""
        from .text_encoder import resize_picture

        extra = req.extra or {}
        if extra.get("prompt_image") is not None:
            return extra["prompt_image"]
        pics = extra.get("image")
        if pics is None:
            return None
        if not isinstance(pics, (list, tuple)):
            pics = [pics]
        out = []
        for im in pics:
            w, h = _picture_size(im)
            cw, ch, _ = calculate_dimensions(CONDITION_IMAGE_SIZE, w / h)
            out.append(resize_picture(im, ch, cw))
        return out

    def resolve_request(self, req: OmniDiffusionRequest, index: int = 0) -> list[dict]:
        extra = req.extra or {}
        images, lat_list = extra.get("image"), extra.get("image_latents")
        if isinstance(images, torch.Tensor):
            images = [images]
        if lat_list is None and not images:
            raise ValueError("the Edit-Plus pipeline needs req.extra['image'] (a list of images) or 'image_latents'")
        # one sample set from the text-to-image resolver, then the condition rows and grids of ALL images
        pics = self._prompt_pictures(req)
        self._prompt_images = [pics] if pics is not None and not isinstance(pics, (list, tuple)) else pics
        try:
            samples = super(QwenImageEditPipeline, self).resolve_request(req, index)
        finally:
            self._prompt_images = None
        packed, grids = [], []
        if lat_list is not None:
            if isinstance(lat_list, torch.Tensor):
                lat_list = [lat_list]
            shapes = extra.get("image_latent_grid")
            if shapes is None or len(shapes) != len(lat_list):
                raise ValueError("image_latents need one `image_latent_grid` entry (h/16, w/16) per condition image")
            if len(shapes) == 2 and isinstance(shapes[0], int):
                shapes = [shapes]
            for z, (gh, gw) in zip(lat_list, shapes):
                packed.append(z.reshape(-1, 64).to(self.device, BF16))
                grids.append((1, int(gh), int(gw)))
        else:
            for im in images:                                               # each image at its own size (:440-462)
                z = self._encode_vae_image(im)
                _, Cz, _, hc, wc = z.shape
                packed.append(self._pack_latents(z[:, :, 0], 1, Cz, hc, wc)[0])
                grids.append((1, hc // 2, wc // 2))
        for p, gr in zip(packed, grids):
            if p.shape[0] != gr[1] * gr[2]:
                raise ValueError("condition-image latents do not match their token grid")
        cond = torch.cat(packed)                                            # [sum S_c, 64]  (:464)
        for sm in samples:
            sm["cond"] = cond
            sm["grid"] = (sm["grid"], *grids)                               # img_shapes = [[(1,h,w), (1,h1,w1), (1,h2,w2), ...]]
        return samples
