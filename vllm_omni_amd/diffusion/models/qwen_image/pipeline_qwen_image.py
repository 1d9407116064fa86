"""QwenImagePipeline on the CDNA4 kernels: latents -> timesteps -> diffuse() -> VAE decode.

Mirror of the reference pipeline (vllm_omni/diffusion/models/qwen_image/pipeline_qwen_image.py:235-754): same class
contract (`__init__(*, od_config, prefix="")`, `forward(req) -> DiffusionOutput`, attributes `.transformer`, `.vae`,
`load_weights`), same helpers (`_pack_latents`, `_unpack_latents`, `prepare_latents`, `prepare_timesteps`, `diffuse`).

What is new relative to the reference (SURVEY.md F6/F7 — neither exists there):
  * the two true-CFG branches of a step run as ONE ragged DiT forward (two items sharing a timestep row) instead
    of two sequential forwards (:556-579);
  * `generate(requests)` step-batches several requests that share (H, W, steps): every denoising step is one DiT
    forward over all their items, with per-request B=1 semantics (no padding, no cross-request attention);
  * CFG combine + norm rescale + Euler update (:580-585) are one fused kernel.

Prompt encoding (Qwen2.5-VL through HF transformers, :357-433; text_encoder.py) is the request-side boundary (SURVEY.md
§8f N1): a request carries either `prompt` strings (needs `text_encoder`) or pre-computed `prompt_embeds`.
"""
from __future__ import annotations

from collections.abc import Iterable

import torch
import torch.nn as nn

from .... import ops
from ...batch import build_ragged_batch
from ...data import DiffusionOutput, OmniDiffusionConfig
from ...request import OmniDiffusionRequest
from .autoencoder_kl_qwenimage import AutoencoderKLQwenImage
from .qwen_image_transformer import QwenImageTransformer2DModel
from .scheduling_flow_match import FlowMatchEulerSchedule

BF16 = torch.bfloat16


def get_qwen_image_post_process_func(od_config: OmniDiffusionConfig):
    """VaeImageProcessor.postprocess(output_type="pil") equivalent (reference :41-60): [-1, 1] float images [B, 3, H, W]
    -> list of PIL images (uint8 HWC arrays with od_config.output_type == "np")."""

    def post_process_func(images: torch.Tensor):
        x = (images.float() / 2 + 0.5).clamp(0, 1)
        arr = (x.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).cpu().numpy()
        if getattr(od_config, "output_type", "pil") != "pil":
            return arr
        from PIL import Image

        return [Image.fromarray(a) for a in arr]

    return post_process_func


class QwenImagePipeline(nn.Module):
    def __init__(self, *, od_config: OmniDiffusionConfig | None = None, prefix: str = "", device=None,
                 transformer: QwenImageTransformer2DModel | None = None, vae: AutoencoderKLQwenImage | None = None,
                 transformer_kwargs: dict | None = None, text_encoder=None):
        super().__init__()
        self.od_config = od_config or OmniDiffusionConfig()
        dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.device = dev
        self.transformer = transformer if transformer is not None else QwenImageTransformer2DModel(
            od_config=self.od_config, device=dev, **(transformer_kwargs or {}))
        self.vae = vae if vae is not None else AutoencoderKLQwenImage(device=dev)
        self.scheduler = FlowMatchEulerSchedule()
        self.text_encoder = text_encoder      # QwenPromptEncoder (text_encoder.py) or None: requests then carry prompt_embeds
        self.vae_scale_factor = 8
        self.default_sample_size = 128
        self._latents_mean = torch.tensor(self.vae.config.latents_mean).view(1, -1, 1, 1, 1)
        self._latents_std = torch.tensor(self.vae.config.latents_std).view(1, -1, 1, 1, 1)
        self.weights_sources: list = []
        self._step_state: dict = {}     # hipGraph + static buffers per step-batch shape
        self._serve_states: dict = {}   # the same for the continuous step batcher (denoise_one_step)
        # Ulysses sequence parallelism: set by the worker when parallel_config.ulysses_degree > 1 — the
        # process group of this rank's SP group; every rank of the group then runs the SAME requests in lockstep
        self.sp_group, self.sp_degree = None, 1
        self.last_teacache_state = None
        self.cache_backend = None
        name = getattr(self.od_config, "cache_backend", "none")
        if name not in (None, "", "none"):
            from ...cache import get_cache_backend

            self.cache_backend = get_cache_backend(name, getattr(self.od_config, "cache_config", {}) or {})
            self.cache_backend.enable(self)

    # ------------------------------------------------------------------ helpers with the reference's semantics
    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width):
        """[B, C, H, W] -> [B, (H/2)(W/2), 4C]: row-major over (h/2, w/2), channel-major inside the 2x2 patch (:436-441)."""
        x = latents.view(batch_size, num_channels_latents, height // 2, 2, width // 2, 2).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        """[B, S, 4C] -> [B, C, 1, H/8, W/8] (:444-457)."""
        B, _, ch = latents.shape
        h = 2 * (int(height) // (vae_scale_factor * 2))
        w = 2 * (int(width) // (vae_scale_factor * 2))
        x = latents.view(B, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
        return x.reshape(B, ch // 4, 1, h, w)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        """(:459-490).  Noise is drawn with the caller's generator; parity runs inject `latents` instead."""
        h = 2 * (int(height) // (self.vae_scale_factor * 2))
        w = 2 * (int(width) // (self.vae_scale_factor * 2))
        if latents is not None:
            return latents.to(device=device, dtype=dtype)
        gdev = generator.device if generator is not None else device
        x = torch.randn((batch_size, 1, num_channels_latents, h, w), generator=generator, device=gdev, dtype=dtype).to(device)
        return self._pack_latents(x, batch_size, num_channels_latents, h, w)

    def prepare_timesteps(self, num_inference_steps, sigmas, image_seq_len):
        """(:492-508) -> (timesteps fp32 [N], N)."""
        ts = self.scheduler.set_timesteps(num_inference_steps, image_seq_len, sigmas)
        return ts, len(ts)

    # ------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def diffuse(self, prompt_embeds, prompt_embeds_mask, negative_prompt_embeds, negative_prompt_embeds_mask, latents,
                img_shapes, txt_seq_lens, negative_txt_seq_lens, timesteps, do_true_cfg, guidance, true_cfg_scale):
        """Reference signature (:530-586) for ONE request batch of B items with equal T; delegates to the
        step-batched core with one 'request' per batch item."""
        B = latents.shape[0]
        shp = img_shapes[0][0] if isinstance(img_shapes[0], (list, tuple)) and isinstance(img_shapes[0][0], (list, tuple)) else img_shapes[0]
        grid = tuple(int(v) for v in shp)
        pos = [prompt_embeds[i, : int(txt_seq_lens[i])] if txt_seq_lens else prompt_embeds[i] for i in range(B)]
        neg = None
        if do_true_cfg:
            neg = [negative_prompt_embeds[i, : int(negative_txt_seq_lens[i])] if negative_txt_seq_lens
                   else negative_prompt_embeds[i] for i in range(B)]
        out = self._denoise(list(latents.unbind(0)), pos, neg, grid, timesteps, self.scheduler.dt(),
                            [float(true_cfg_scale)] * B)
        return torch.stack(out)

    @torch.no_grad()
    def _denoise(self, latents: list[torch.Tensor], pos: list[torch.Tensor], neg: list[torch.Tensor] | None,
                 grid, timesteps: torch.Tensor, dts: torch.Tensor, cfg_scales: list[float],
                 cond: list[torch.Tensor] | None = None, cfg_normalize: bool = True, t_cond: int | None = None) -> list[torch.Tensor]:
        """Step-batched denoising of R requests sharing (grid, schedule).  latents[r] [S_img, 64];
        pos[r]/neg[r] [T, joint_dim] (ragged T).  Item order: pos_0..pos_{R-1}, then neg_0..neg_{R-1}.

        One step = copy latents into the forward's input rows -> ONE ragged DiT forward over all items -> fused
        CFG-combine + norm-rescale + Euler update.  When `use_hip_graph` applies, that step (~670 launches at 60 layers) is
        captured once per (batch shape, cfg) as a hipGraph and replayed: at 256^2 a forward is ~4 ms of GPU work against
        ~2.5 ms of host launch time (SURVEY.md §7 'Hard parts').

        `cond[r]` [S_c, 64] (Edit pipelines): packed condition-image latents appended to request r's rows on the sequence
        axis in every forward and sliced off the prediction (pipeline_qwen_image_edit.py:600-632); `grid` is then the
        sequence of token grids ((1, h, w), (1, h_c, w_c), ...).  `cfg_normalize` = False: true-CFG combination without the norm
        rescale; `t_cond`: the Layered variant's `additional_t_cond` (is_rgb) for every item."""
        tr, dev = self.transformer, self.device
        if self.sp_degree > 1 or getattr(self, "_force_sp_path", False) or getattr(self, "_sp_emulate_ranks", 0):
            return self._denoise_sp(latents, pos, neg, grid, timesteps, dts, cfg_scales, cond, cfg_normalize, t_cond)
        R = len(latents)
        S = latents[0].shape[0]
        S_c = 0 if cond is None else int(cond[0].shape[0])
        S_tot = S + S_c
        do_cfg = neg is not None
        if do_cfg and len(set(cfg_scales)) != 1:
            raise NotImplementedError("step-batched requests must share true_cfg_scale")
        txt = [p.to(dev, BF16) for p in pos] + ([n.to(dev, BF16) for n in neg] if do_cfg else [])
        lens = [int(t.shape[0]) for t in txt]
        # all requests are at the same timestep: ONE temb row, shared by every item (both CFG branches included)
        rb = build_ragged_batch(lens, grid, temb_rows=[0] * len(lens))
        prepared = tr.prepare_batch(rb)
        n_items = (2 if do_cfg else 1) * R
        sig_host = self.scheduler.model_timestep(timesteps)                         # bf16-rounded t/1000, fp32 [N]
        sig_in = sig_host.to(dev)
        dt_dev = dts.to(dev, torch.float32).contiguous()
        graph_on = self._use_graph(n_items * S)
        tcfg = getattr(tr, "teacache", None)
        key = (tuple(lens), tuple(grid), do_cfg, float(cfg_scales[0]), R, bool(cfg_normalize), t_cond,
               bool(getattr(self.od_config, "precompute_modulation", True)),
               None if tcfg is None else (tcfg.rel_l1_thresh, tuple(tcfg.coefficients)))
        st = self._step_state.get(key) if graph_on else None
        if st is None:
            st = dict(lat=torch.empty(R * S, tr.in_channels, dtype=BF16, device=dev),
                      lat_in=torch.empty(n_items * S_tot, tr.in_channels, dtype=BF16, device=dev),
                      pred=torch.empty(n_items * S_tot, tr.in_channels, dtype=BF16, device=dev),
                      pred_c=torch.empty(n_items * S, tr.in_channels, dtype=BF16, device=dev) if S_c else None,
                      prompt=torch.empty(sum(lens), tr.joint_attention_dim, dtype=BF16, device=dev),
                      sig=torch.empty(1, dtype=torch.float32, device=dev), dt=torch.empty(1, dtype=torch.float32, device=dev),
                      graph=None, tc=None)
            if tcfg is not None:
                from ...cache.teacache.native import TeaCacheDeviceState

                st["tc"] = TeaCacheDeviceState(tcfg, rb, tr.inner_dim, dev)
            if graph_on:
                if len(self._step_state) >= 8:
                    self._step_state.clear()
                self._step_state[key] = st
        st["lat"].copy_(torch.cat([x.to(dev, BF16) for x in latents]))
        st["prompt"].copy_(torch.cat(txt))
        lat, lat_in, pred = st["lat"], st["lat_in"], st["pred"]
        Cl = tr.in_channels
        if S_c:                                              # the condition rows never change: written once per loop
            cv = torch.stack([c.to(dev, BF16) for c in cond])                       # [R, S_c, 64]
            li = lat_in.view(n_items, S_tot, Cl)
            li[:R, S:].copy_(cv)
            if do_cfg:
                li[R:, S:].copy_(cv)
        tr.do_true_cfg = do_cfg
        tc = st["tc"]
        if tc is not None:
            tc.reset()                                       # a new generation: first forward always computes
        # Layered variant: the addition_t_embedding rows of this batch, gathered once, outside any graph capture
        temb_add = None
        if t_cond is not None:
            rows = tr.time_text_embed.additional_rows([t_cond] * rb.n_temb, rb.n_temb)
            if st.get("temb_add") is None:
                st["temb_add"] = rows                        # kept with the step state: a captured graph holds its address ...
            else:
                st["temb_add"].copy_(rows)                   # ... so later generations refresh the VALUES in place
            temb_add = st["temb_add"]

        # modulation vectors of every step of this schedule, computed once (all requests of a static step-batch share it)
        mod_tab = None
        if getattr(self.od_config, "precompute_modulation", True):
            mod_tab = tr.modulation_table_for_schedule(sig_in, None if t_cond is None else [t_cond] * len(timesteps),
                                                       cache=bool(getattr(self.od_config, "cache_modulation_tables", True)),
                                                       sigma_host=sig_host)                                # [L, 2, N, 6D]
            if st.get("mod") is None:
                st["mod"] = torch.empty(mod_tab.shape[0], 2, 1, mod_tab.shape[3], dtype=BF16, device=dev)  # static: graphs bake it
        mod_now = st.get("mod") if mod_tab is not None else None
        if mod_now is not None:
            mod_now.copy_(mod_tab[:, :, 0:1])                 # defined contents for a graph's warm-up / capture forwards

        def step(sig1, dt1):
            if S_c:
                li = lat_in.view(n_items, S_tot, Cl)
                li[:R, :S].copy_(lat.view(R, S, Cl))
                if do_cfg:
                    li[R:, :S].copy_(lat.view(R, S, Cl))
            else:
                lat_in[: R * S].copy_(lat)
                if do_cfg:
                    lat_in[R * S:].copy_(lat)
            tr.forward_ragged(prepared, lat_in, st["prompt"], sig1, out=pred, teacache=tc, temb_add=temb_add, mod_table=mod_now)
            pr = pred
            if S_c:                                          # noise_pred[:, :latents.size(1)] (edit pipeline :632)
                st["pred_c"].view(n_items, S, Cl).copy_(pred.view(n_items, S_tot, Cl)[:, :S])
                pr = st["pred_c"]
            ops.cfg_euler_step_(lat, pr[: R * S], pr[R * S:] if do_cfg else None, cfg_scales[0], dt1, normalize=cfg_normalize)

        self.last_teacache_state = tc                         # statistics: tc.skipped_forwards() per item after the loop
        if not graph_on:
            for i in range(len(timesteps)):
                if mod_now is not None:
                    mod_now.copy_(mod_tab[:, :, i:i + 1])
                step(sig_in[i:i + 1], dt_dev[i:i + 1])
            return list(lat.clone().view(R, S, -1).unbind(0))
        tr._native_weights()      # (re)build the pointer table NOW: `_native_gen` then names the storages a replay would read
        if st["graph"] is None or st.get("gen") != tr._native_gen:
            # warm-up on a side stream (lazy initialisation inside the native library: function attributes, workspace),
            # then capture; the warm-up step advances `lat`, so the real inputs are restored afterwards
            saved = lat.clone()
            st["sig"].copy_(sig_in[:1]); st["dt"].zero_()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                step(st["sig"], st["dt"])
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step(st["sig"], st["dt"])
            st["graph"] = g
            if tc is not None:
                tc.reset()                                   # the warm-up / capture forwards advanced the counters
            # everything whose device address is baked into the captured kernel arguments stays referenced by the entry
            st["keepalive"], st["gen"] = (prepared, tr._workspace, tr._native), tr._native_gen
            lat.copy_(saved)
        for i in range(len(timesteps)):
            st["sig"].copy_(sig_in[i:i + 1], non_blocking=True)
            st["dt"].copy_(dt_dev[i:i + 1], non_blocking=True)
            if mod_now is not None:
                mod_now.copy_(mod_tab[:, :, i:i + 1])
            st["graph"].replay()
        return list(lat.clone().view(R, S, -1).unbind(0))

    @torch.no_grad()
    def _denoise_sp(self, latents, pos, neg, grid, timesteps, dts, cfg_scales, cond=None, cfg_normalize: bool = True,
                    t_cond: int | None = None) -> list[torch.Tensor]:
        """The same loop with every DiT forward SEQUENCE-PARALLEL over `self.sp_group` (reference wiring:
        qwen_image_transformer.py:735-742,776-781,800-801 + attention/parallel/ulysses.py:59-135; end-to-end contract
        tests/e2e/offline_inference/test_sequence_parallel.py:68-71,128-147): each rank holds S / P image rows of an
        item through the block stack, attention runs on H / P heads over the whole sequence.  The items of a step (requests
        x CFG branches) are independent forwards: they are software-pipelined so that one item's all-to-all flies while the
        next item's GEMMs run (distributed/sp_driver.py).  Every rank ends with the full latents.

        The strategy is pipeline-agnostic, as the reference's is (its Ulysses layer wraps any DiT forward):
          * `cond` (Edit / Edit-Plus / Layered): the condition-image rows ride on the sequence axis of every forward — the rows
            [latents ; condition] are sharded as ONE sequence (the reference chunks `hidden_states` after the pipeline's
            `torch.cat([latents, image_latents], dim=1)`, pipeline_qwen_image_edit.py:600-632) and the prediction is cut back to
            the latent rows; `grid` is then the sequence of token grids;
          * `t_cond` / `cfg_normalize` (Layered): the additional_t_cond rows join the timestep embedding on every rank, the
            true-CFG combination runs without the norm rescale;
          * TeaCache: one TeaCacheSPState per item and rank — per-rank residual slices, ONE all-reduced pair of sums per forward
            for the decision, so every rank takes the single-device decision (cache/teacache/sp_state.py)."""
        tr, dev = self.transformer, self.device
        do_cfg = neg is not None
        if do_cfg and len(set(cfg_scales)) != 1:
            raise NotImplementedError("step-batched requests must share true_cfg_scale")
        R, S = len(latents), int(latents[0].shape[0])
        lat = torch.stack([x.to(dev, BF16) for x in latents]).contiguous()                  # [R, S, 64]
        pe = [p.to(dev, BF16) for p in pos] + ([n.to(dev, BF16) for n in neg] if do_cfg else [])
        cv = None if cond is None else [c.to(dev, BF16) for c in cond]                      # [S_c, 64] per request
        sig_in = self.scheduler.model_timestep(timesteps).to(dev)
        dt_dev = dts.to(dev, torch.float32).contiguous()
        tr.do_true_cfg = do_cfg
        flat = lat.view(R * S, -1)
        n_items = (2 if do_cfg else 1) * R
        emu = int(getattr(self, "_sp_emulate_ranks", 0) or 0)
        tcfg = getattr(tr, "teacache", None)
        tc_states = None
        self.last_teacache_state = None
        if tcfg is not None:
            from ...cache.teacache.sp_state import TeaCacheSPState, TeaCacheSPStats

            tc_states = ([[TeaCacheSPState(tcfg) for _ in range(n_items)] for _ in range(emu)] if emu
                         else [TeaCacheSPState(tcfg) for _ in range(n_items)])
            self.last_teacache_state = TeaCacheSPStats(tc_states[0] if emu else tc_states)
        for i in range(len(timesteps)):
            sg = sig_in[i:i + 1]
            rows = [lat[r] if cv is None else torch.cat([lat[r], cv[r]]) for r in range(R)]
            items = [(rows[r % R], pe[j], sg) for j, r in enumerate(list(range(R)) * (2 if do_cfg else 1))]
            preds = tr.forward_sp_multi(items, grid, self.sp_group, t_cond=t_cond, teacache_states=tc_states, emulate_ranks=emu)
            preds = [p[:S] for p in preds]                                                  # noise_pred[:, :latents.size(1)]
            p = torch.cat(preds[:R]).contiguous()
            n = torch.cat(preds[R:]).contiguous() if do_cfg else None
            ops.cfg_euler_step_(flat, p, n, cfg_scales[0], dt_dev[i:i + 1], normalize=cfg_normalize)
        return list(lat.clone().unbind(0))

    def _use_graph(self, img_rows: int) -> bool:
        """`od_config.use_hip_graph`: True / False, or None = automatic (on while a forward is short enough for the host's
        launch rate to matter: <= 4096 image rows, i.e. up to 512^2 x 4 items or one 1024^2 item)."""
        flag = getattr(self.od_config, "use_hip_graph", None)
        if flag is None:
            return img_rows <= 4096
        return bool(flag)

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def decode_latents(self, latents: torch.Tensor, height: int, width: int) -> torch.Tensor:
        """packed latents [B, S, 64] -> image [B, 3, H, W] in [-1, 1]  (:736-747)."""
        z = self._unpack_latents(latents, height, width, self.vae_scale_factor).to(self.vae.dtype)
        mean = self._latents_mean.to(z.device, z.dtype)
        inv_std = 1.0 / self._latents_std.to(z.device, z.dtype)
        z = z / inv_std + mean
        return self.vae.decode(z, return_dict=False)[0][:, :, 0]

    # ------------------------------------------------------------------ request level
    def _req_params(self, req: OmniDiffusionRequest):
        """(height, width, steps, cfg scale, true-CFG on?) with the reference's defaults (:614-624, :655-659): the negative
        prompt defaults to "" — NOT None — so true-CFG is ON for a plain text request.  Raises the reference's
        check_inputs errors (:288-349) for inconsistent requests."""
        height = req.height or self.default_sample_size * self.vae_scale_factor
        width = req.width or self.default_sample_size * self.vae_scale_factor
        steps = req.num_inference_steps or 50
        cfg = req.true_cfg_scale or 4.0
        # A request that carries BOTH a prompt string and pre-computed embeddings is served from the embeddings (the prompt
        # is then metadata for the output record): the entry point `OmniDiffusion.generate(prompt, prompt_embeds=...)` always
        # sets `prompt`, so the reference's either-or check (:300-307) would make embeddings unusable through it.
        if req.prompt is None and req.prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if req.prompt is not None and not isinstance(req.prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(req.prompt)}")
        if req.negative_prompt is not None and req.negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if req.prompt is not None and req.prompt_embeds is None and self.text_encoder is None:
            raise NotImplementedError("this pipeline was built without a text encoder: pass prompt_embeds, construct it with "
                                      "text_encoder=, or call load_text_encoder(checkpoint_dir)")
        if (req.num_outputs_per_prompt or 1) < 1:
            raise ValueError("num_outputs_per_prompt must be >= 1")
        if req.prompt is not None and req.prompt_embeds is None:
            has_neg = True                       # negative_prompt defaults to "" in the reference's forward (:591-592)
        else:
            has_neg = req.negative_prompt_embeds is not None or req.negative_prompt is not None
        return height, width, steps, cfg, (cfg > 1 and has_neg)

    def load_text_encoder(self, model_dir: str, device=None) -> None:
        """Build the prompt encoder from the `text_encoder/` + `tokenizer/` folders of a diffusers-layout checkpoint
        (reference :225-228,261: `Qwen2_5_VLForConditionalGeneration.from_pretrained(model, subfolder="text_encoder")`,
        `Qwen2Tokenizer.from_pretrained(model, subfolder="tokenizer")`), local files only."""
        import os

        from transformers import AutoTokenizer, Qwen2_5_VLForConditionalGeneration

        from .text_encoder import QwenPromptEncoder

        te_dir, tok_dir = os.path.join(model_dir, "text_encoder"), os.path.join(model_dir, "tokenizer")
        if not (os.path.isdir(te_dir) and os.path.isdir(tok_dir)):
            raise FileNotFoundError(f"{model_dir!r} has no text_encoder/ + tokenizer/ folders")
        dev = torch.device(device if device is not None else self.device)
        model = Qwen2_5_VLForConditionalGeneration.from_pretrained(te_dir, torch_dtype=BF16, local_files_only=True).to(dev).eval()
        tok = AutoTokenizer.from_pretrained(tok_dir, local_files_only=True)
        self.text_encoder = QwenPromptEncoder(model, tok, dtype=BF16)

    def encode_prompt(self, prompt, num_images_per_prompt: int = 1, prompt_embeds=None, prompt_embeds_mask=None,
                      max_sequence_length: int = 1024):
        """Reference signature (:394-433): -> (prompt_embeds [B*n, T, 3584] zero-padded, mask [B*n, T])."""
        prompt = [prompt] if isinstance(prompt, str) else prompt
        if prompt_embeds is None:
            prompt_embeds, prompt_embeds_mask = self.text_encoder.get_qwen_prompt_embeds(prompt, device=self.device)
        B = prompt_embeds.shape[0]
        prompt_embeds, prompt_embeds_mask = prompt_embeds[:, :max_sequence_length], prompt_embeds_mask[:, :max_sequence_length]
        T = prompt_embeds.shape[1]
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(B * num_images_per_prompt, T, -1)
        prompt_embeds_mask = prompt_embeds_mask.repeat(1, num_images_per_prompt, 1).view(B * num_images_per_prompt, T)
        return prompt_embeds, prompt_embeds_mask

    def _encode_text(self, prompts: list[str]):
        """Prompt strings -> (embeds, mask).  The Edit pipelines override this to show the picture(s) to the vision tower."""
        return self.text_encoder.get_qwen_prompt_embeds(prompts, device=self.device)

    def _rows_of(self, embeds, mask, prompts, n_per_prompt: int) -> list[torch.Tensor]:
        """Per-sample [T_i, joint] rows (padding removed) from either pre-computed embeddings (+ mask) or prompt strings."""
        if embeds is None:
            embeds, mask = self._encode_text([prompts] if isinstance(prompts, str) else list(prompts))
        if embeds.dim() == 2:
            embeds = embeds.unsqueeze(0)
        if mask is not None:
            mask = mask.reshape(embeds.shape[0], -1)         # [T] with [T, D] / [1, T, D] embeds -> [1, T]
            if mask.shape[1] != embeds.shape[1]:
                raise ValueError(f"prompt_embeds_mask {tuple(mask.shape)} does not match prompt_embeds {tuple(embeds.shape)}")
        rows = []
        for b in range(embeds.shape[0]):
            t = int(mask[b].sum()) if mask is not None else embeds.shape[1]
            rows += [embeds[b, :t]] * n_per_prompt
        return rows

    def resolve_request(self, req: OmniDiffusionRequest, index: int = 0) -> list[dict]:
        """Expand one request into its SAMPLES (prompts x num_outputs_per_prompt, the reference's
        `batch_size * num_images_per_prompt`, :663-683): each sample is an independent denoising problem with B=1 semantics."""
        height, width, steps, cfg, do_cfg = self._req_params(req)
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
        gh, gw = height // self.vae_scale_factor // 2, width // self.vae_scale_factor // 2
        S = gh * gw
        lat_all = None
        if req.latents is not None:
            lat_all = req.latents.reshape(-1, S, req.latents.shape[-1])
            if lat_all.shape[0] not in (1, len(pos)):
                raise ValueError(f"latents carry {lat_all.shape[0]} samples, the request expands to {len(pos)}")
        gen = req.generator
        if gen is None and req.seed is not None:
            gen = torch.Generator(device="cpu").manual_seed(req.seed)
        samples = []
        for k in range(len(pos)):
            if lat_all is not None:
                lat = lat_all[k if lat_all.shape[0] > 1 else 0].to(self.device, BF16)
            else:
                g = gen[k] if isinstance(gen, (list, tuple)) else gen
                lat = self.prepare_latents(1, self.transformer.in_channels // 4, height, width, BF16, self.device, g)[0]
            samples.append(dict(req=index, k=k, height=height, width=width, steps=steps, cfg=float(cfg), do_cfg=do_cfg,
                                grid=(1, gh, gw), lat=lat, pos=pos[k], neg=neg[k] if do_cfg else None))
        return samples

    @torch.no_grad()
    def generate(self, requests: list[OmniDiffusionRequest], output_type: str = "pt") -> list[DiffusionOutput]:
        """Run requests; samples that share (height, width, steps, cfg on/off, cfg scale) are step-batched, up to
        `od_config.max_step_batch` per DiT forward.  A request with several prompts / num_outputs_per_prompt > 1 returns
        its samples stacked along dim 0."""
        samples = [s for i, r in enumerate(requests) for s in self.resolve_request(r, i)]
        groups: dict[tuple, list[int]] = {}
        for j, sm in enumerate(samples):
            groups.setdefault((sm["height"], sm["width"], sm["steps"], sm["cfg"], sm["do_cfg"], sm["grid"], sm.get("mu"),
                               sm.get("cfg_normalize", True), sm.get("t_cond")), []).append(j)
        cap = max(1, int(getattr(self.od_config, "max_step_batch", 4)))
        final: list[torch.Tensor | None] = [None] * len(samples)
        for (height, width, steps, cfg, do_cfg, _grid, mu, cfg_norm, t_cond), idxs in groups.items():
            for s0 in range(0, len(idxs), cap):
                chunk = [samples[j] for j in idxs[s0:s0 + cap]]
                timesteps = self.scheduler.set_timesteps(steps, chunk[0]["lat"].shape[0], chunk[0].get("sigmas"), mu=mu)
                outs = self._denoise([c["lat"] for c in chunk], [c["pos"] for c in chunk],
                                     [c["neg"] for c in chunk] if do_cfg else None, chunk[0]["grid"], timesteps,
                                     self.scheduler.dt(), [cfg] * len(chunk),
                                     cond=[c["cond"] for c in chunk] if chunk[0].get("cond") is not None else None,
                                     cfg_normalize=cfg_norm, t_cond=t_cond)
                for j, o in zip(idxs[s0:s0 + cap], outs):
                    final[j] = o
        # decode: the images of ALL requests of one size in one VAE call (the decoder's small rasters — 128^2 and 256^2 at 1024^2
        # — fill a quarter of the chip per image; per image a batch of five decodes 15 % faster than five calls).  The conv
        # launcher picks its tile shape and norm fusion from the TOTAL tile count of a call (csrc/vae.hip conv_uses_big_tile),
        # so an image decoded in a batch and the same image decoded alone agree to bf16 rounding of another summation order
        # (<= 1e-2 rel_l2, tests/test_gpu_vae_ops.py), not bit for bit, wherever the batch flips that choice
        mine = [[j for j, sm in enumerate(samples) if sm["req"] == i] for i in range(len(requests))]
        want = [not (output_type == "latent" or r.output_type == "latent") for r in requests]
        lats = [torch.stack([final[j] for j in m]) for m in mine]                 # per request [n_samples, S, 64]
        images: list[torch.Tensor | None] = [None] * len(requests)
        by_size: dict[tuple, list[int]] = {}
        for i, m in enumerate(mine):
            if want[i]:
                by_size.setdefault((samples[m[0]]["height"], samples[m[0]]["width"], lats[i].shape[1]), []).append(i)
        for _size, reqs in by_size.items():
            # chunks of at most DECODE_BATCH LATENTS (a request with num_outputs_per_prompt > 1 carries several: chunking by
            # request would exceed the activation bound the constant stands for)
            flat = torch.cat([lats[i] for i in reqs])                                # [n_latents, S, 64], request order
            sample0 = samples[mine[reqs[0]][0]]
            dec = torch.cat([self._decode_samples(flat[c0:c0 + self.DECODE_BATCH], sample0)
                             for c0 in range(0, flat.shape[0], self.DECODE_BATCH)])
            per = dec.shape[0] // flat.shape[0]                                      # images per latent (Layered: one per layer)
            o = 0
            for i in reqs:
                n = lats[i].shape[0] * per
                images[i] = dec[o:o + n]
                o += n
        return [DiffusionOutput(output=images[i] if want[i] else lats[i]) for i in range(len(requests))]

    DECODE_BATCH = 8                                                 # latents per VAE call (activations: 1 GB per latent at 1024^2)

    def _decode_samples(self, lat: torch.Tensor, sample: dict) -> torch.Tensor:
        """Finished packed latents [n, S, 64] of one size -> images (the Layered pipeline decodes one image per layer)."""
        return self.decode_latents(lat, sample["height"], sample["width"])

    # ------------------------------------------------------------------ continuous step batching (step_batcher.py)
    def begin_sample(self, a) -> None:
        """Schedule of one sample: per-step model timestep (bf16-rounded t/1000) and dt as DEVICE vectors.  A step reads its
        values from per-slot device tables with a device-side step counter: nothing a step needs lives in host memory, so the
        host may enqueue many steps ahead of the GPU (it does: ~10 ms to enqueue a 540-ms step) without racing a pending
        host-to-device copy."""
        sm = a.sample
        sch = FlowMatchEulerSchedule(self.scheduler.config)
        ts = sch.set_timesteps(sm["steps"], sm["lat"].shape[0], sm.get("sigmas"), mu=sm.get("mu"))
        a.n_steps = len(ts)
        if a.n_steps >= self.SERVE_MAX_STEPS:
            raise NotImplementedError(f"{a.n_steps} denoising steps: the step batcher's schedule tables hold {self.SERVE_MAX_STEPS - 1}")
        sig_h = sch.model_timestep(ts).float()
        sig_d = sig_h.to(self.device)
        # mod_tab: this sample's modulation table [L, 2, n_steps, 6D] (88 MB at 20 steps) — built when the sample first takes
        # a slot of a running batch, not at admission: a long queue of waiting requests must not hold one each
        a.state = dict(sig_d=sig_d, sig_h=sig_h, dt_d=sch.dt().float().to(self.device), mod_tab=None,
                       lat=sm["lat"].to(self.device, BF16).clone(), pos=sm["pos"].to(self.device, BF16),
                       neg=None if sm["neg"] is None else sm["neg"].to(self.device, BF16),
                       cond=None if sm.get("cond") is None else sm["cond"].to(self.device, BF16),
                       home=None, tc=None)

    SERVE_MAX_STEPS = 1024

    @staticmethod
    def batch_key(a):
        sm = a.sample
        return (sm["grid"], sm["do_cfg"], sm["cfg"], sm.get("cfg_normalize", True), sm.get("t_cond"))

    # A "serve state" = everything one composition of a running batch needs, allocated ONCE: the static input / output
    # buffers of the forward, the ragged-batch descriptor, the per-request timestep / dt vectors, the TeaCache device state
    # and (for short forwards) the captured hipGraph of one step.  A sample lives in one slot of one state ("home"): its
    # latents and its TeaCache history stay there from step to step, and are copied out / in only when the batch is
    # re-composed (a request joined or left) — never per step.  Item order: pos_0..pos_{R-1}, neg_0..neg_{R-1}.
    def _serve_state(self, group: list) -> dict:
        tr, dev = self.transformer, self.device
        sm0 = group[0].sample
        R, do_cfg = len(group), sm0["do_cfg"]
        S = int(group[0].state["lat"].shape[0])
        cond0 = group[0].state.get("cond")
        S_c = 0 if cond0 is None else int(cond0.shape[0])
        lens = [int(a.state["pos"].shape[0]) for a in group] + ([int(a.state["neg"].shape[0]) for a in group] if do_cfg else [])
        tcfg = getattr(tr, "teacache", None)
        use_tab = bool(getattr(self.od_config, "precompute_modulation", True))
        key = (tuple(lens), tuple(sm0["grid"]), do_cfg, float(sm0["cfg"]), R, S_c, sm0.get("cfg_normalize", True), sm0.get("t_cond"),
               use_tab, None if tcfg is None else (tcfg.rel_l1_thresh, tuple(tcfg.coefficients)))
        st = self._serve_states.get(key)
        if st is not None:
            return st
        n_items, S_tot, Cl = (2 if do_cfg else 1) * R, S + S_c, tr.in_channels
        rb = build_ragged_batch(lens, sm0["grid"], temb_rows=list(range(R)) * (2 if do_cfg else 1))
        offs = [0]
        for t in lens:
            offs.append(offs[-1] + t)
        st = dict(key=key, R=R, S=S, S_c=S_c, do_cfg=do_cfg, cfg=float(sm0["cfg"]), n_items=n_items, rb=rb, txt_off=offs,
                  cfg_normalize=bool(sm0.get("cfg_normalize", True)), t_cond=sm0.get("t_cond"),
                  temb_add=None if sm0.get("t_cond") is None else tr.time_text_embed.additional_rows([sm0["t_cond"]] * R, R),
                  prepared=tr.prepare_batch(rb), members=[None] * R, graph=None, gen=None,
                  lat=torch.zeros(R * S, Cl, dtype=BF16, device=dev),
                  lat_in=torch.zeros(n_items * S_tot, Cl, dtype=BF16, device=dev),
                  pred=torch.empty(n_items * S_tot, Cl, dtype=BF16, device=dev),
                  pred_c=torch.empty(n_items * S, Cl, dtype=BF16, device=dev) if S_c else None,
                  prompt=torch.zeros(sum(lens), tr.joint_attention_dim, dtype=BF16, device=dev),
                  sig=torch.zeros(R, dtype=torch.float32, device=dev), dt=torch.zeros(R, dtype=torch.float32, device=dev),
                  # per-slot schedules and step counters, all on the device (see begin_sample)
                  sig_tab=torch.zeros(R, self.SERVE_MAX_STEPS, dtype=torch.float32, device=dev),
                  dt_tab=torch.zeros(R, self.SERVE_MAX_STEPS, dtype=torch.float32, device=dev),
                  step_idx=torch.zeros(R, dtype=torch.int64, device=dev), tc=None,
                  # this step's modulation rows of every resident sample (gathered from the samples' tables before each step)
                  mod=(torch.zeros(len(tr.transformer_blocks), 2, R, 6 * tr.inner_dim, dtype=BF16, device=dev) if use_tab else None))
        if tcfg is not None:
            from ...cache.teacache.native import TeaCacheDeviceState

            st["tc"] = TeaCacheDeviceState(tcfg, rb, tr.inner_dim, dev)
        if len(self._serve_states) >= 8:                      # evict: every resident sample goes back to its own tensors
            for old in self._serve_states.values():
                for occ in old["members"]:
                    if occ is not None:
                        self._export_sample(occ)
            self._serve_states.clear()
        self._serve_states[key] = st
        return st

    def _export_sample(self, a) -> None:
        """Copy a sample's latents (and TeaCache history) out of its home slot; the slot becomes free."""
        home = a.state.get("home")
        if home is None:
            return
        st, r = home
        S, R = st["S"], st["R"]
        a.state["lat"] = st["lat"][r * S:(r + 1) * S].clone()
        if st["tc"] is not None:
            a.state["tc"] = [st["tc"].export_item(r)] + ([st["tc"].export_item(R + r)] if st["do_cfg"] else [])
        st["members"][r] = None
        a.state["home"] = None

    def _import_sample(self, a, st: dict, r: int) -> None:
        home = a.state.get("home")
        if home is not None and home[0] is st and home[1] == r:
            return
        self._export_sample(a)
        occ = st["members"][r]
        if occ is not None and occ is not a:
            self._export_sample(occ)
        S, S_c, R, do_cfg = st["S"], st["S_c"], st["R"], st["do_cfg"]
        st["lat"][r * S:(r + 1) * S].copy_(a.state["lat"])
        n = a.state["sig_d"].shape[0]
        st["sig_tab"][r, :n].copy_(a.state["sig_d"])
        st["dt_tab"][r, :n].copy_(a.state["dt_d"])
        st["step_idx"][r:r + 1].fill_(a.step)               # (a kernel argument, not a host buffer read later)
        off = st["txt_off"]
        st["prompt"][off[r]:off[r + 1]].copy_(a.state["pos"])
        if do_cfg:
            st["prompt"][off[R + r]:off[R + r + 1]].copy_(a.state["neg"])
        if S_c:                                              # Edit: the condition rows of this sample's item(s) never change
            li = st["lat_in"].view(st["n_items"], S + S_c, -1)
            li[r, S:].copy_(a.state["cond"])
            if do_cfg:
                li[R + r, S:].copy_(a.state["cond"])
        if st["tc"] is not None:
            saved = a.state.get("tc")
            st["tc"].import_item(r, saved[0] if saved else None)
            if do_cfg:
                st["tc"].import_item(R + r, saved[1] if saved else None)
        st["members"][r] = a
        a.state["home"] = (st, r)

    def _serve_step_body(self, st: dict) -> None:
        tr = self.transformer
        R, S, S_c, do_cfg, n_items = st["R"], st["S"], st["S_c"], st["do_cfg"], st["n_items"]
        lat, lat_in, pred = st["lat"], st["lat_in"], st["pred"]
        Cl = lat.shape[1]
        idx = st["step_idx"].clamp(max=self.SERVE_MAX_STEPS - 1).view(R, 1)
        st["sig"].copy_(st["sig_tab"].gather(1, idx).view(R))
        st["dt"].copy_(st["dt_tab"].gather(1, idx).view(R))
        if S_c:
            li = lat_in.view(n_items, S + S_c, Cl)
            li[:R, :S].copy_(lat.view(R, S, Cl))
            if do_cfg:
                li[R:, :S].copy_(lat.view(R, S, Cl))
        else:
            lat_in[: R * S].copy_(lat)
            if do_cfg:
                lat_in[R * S:].copy_(lat)
        tr.forward_ragged(st["prepared"], lat_in, st["prompt"], st["sig"], out=pred, teacache=st["tc"], temb_add=st["temb_add"],
                          mod_table=st["mod"])
        pr = pred
        if S_c:                                              # noise_pred[:, :latents.size(1)] (edit pipeline :632)
            st["pred_c"].view(n_items, S, Cl).copy_(pred.view(n_items, S + S_c, Cl)[:, :S])
            pr = st["pred_c"]
        ops.cfg_euler_step_(lat, pr[: R * S], pr[R * S:] if do_cfg else None, st["cfg"], st["dt"], dt_rows_per_item=S,
                            normalize=st["cfg_normalize"])
        st["step_idx"].add_(1)

    @torch.no_grad()
    def denoise_one_step(self, group: list) -> None:
        """ONE ragged DiT forward for samples that may sit at different step indices: sample r owns temb row r (its CFG
        pair shares it); the fused CFG + Euler kernel reads a per-sample dt.  Static buffers, no per-step concatenation;
        TeaCache decisions per item on the device; the step is a hipGraph replay when the forward is short (reference loop
        shape: vllm_omni/diffusion/worker/gpu_worker.py:226-290 runs one request to completion per iteration)."""
        tr, dev = self.transformer, self.device
        if self.sp_degree > 1:
            raise NotImplementedError("sequence-parallel workers run requests to completion (generate), not step-batched")
        st = self._serve_state(group)
        for r, a in enumerate(group):
            self._import_sample(a, st, r)
            if st["mod"] is not None:
                if a.state.get("mod_tab") is None:
                    tc_ = a.sample.get("t_cond")
                    a.state["mod_tab"] = tr.modulation_table_for_schedule(
                        a.state["sig_d"], None if tc_ is None else [tc_] * a.n_steps,
                        cache=bool(getattr(self.od_config, "cache_modulation_tables", True)), sigma_host=a.state.get("sig_h"))
                # sample r's modulation rows for ITS step (`a.step`: the host's count of the steps enqueued for it — equal to
                # the device-side counter when this copy executes, and the source table is never written again)
                st["mod"][:, :, r].copy_(a.state["mod_tab"][:, :, min(a.step, a.n_steps - 1)])
        tr.do_true_cfg = st["do_cfg"]
        self.last_teacache_state = st["tc"]
        if not self._use_graph(st["n_items"] * st["S"]):
            self._serve_step_body(st)
            return
        tr._native_weights()
        if st["graph"] is None or st["gen"] != tr._native_gen:
            # warm-up on a side stream, then capture (see _denoise); both advance `lat` and the TeaCache state of the resident
            # samples, so those are saved and restored around them
            saved_lat, saved_idx = st["lat"].clone(), st["step_idx"].clone()
            saved_tc = None
            if st["tc"] is not None:
                saved_tc = [st["tc"].export_item(i) for i in range(st["n_items"])]
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._serve_step_body(st)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._serve_step_body(st)
            st["graph"], st["gen"] = g, tr._native_gen
            st["keepalive"] = (st["prepared"], tr._workspace, tr._native)
            st["lat"].copy_(saved_lat)
            st["step_idx"].copy_(saved_idx)
            if saved_tc is not None:
                for i, sv in enumerate(saved_tc):
                    st["tc"].import_item(i, sv)
        st["graph"].replay()

    def sample_result(self, a) -> torch.Tensor:
        """Final latents of a finished sample; its slot is released."""
        self._export_sample(a)
        return a.state["lat"]

    def finish_request(self, req: OmniDiffusionRequest, latents: list[torch.Tensor], sample: dict) -> DiffusionOutput:
        lat = torch.stack(latents)
        if req.output_type == "latent":
            return DiffusionOutput(output=lat)
        return DiffusionOutput(output=self._decode_samples(lat, sample))

    def forward(self, req: OmniDiffusionRequest, **_kw) -> DiffusionOutput:
        """Reference entry point (:588-750): one request in, DiffusionOutput out."""
        return self.generate([req])[0]

    def expected_weight_names(self) -> set[str]:
        """Checkpoint names `load_weights` must see for a complete load (the loader's "weights not initialized" check,
        reference diffusers_loader.py:247-258): the DiT's parameters under their fused names, the VAE's under the names of
        the diffusers / vendored state dict."""
        names = {"transformer." + n for n, _ in self.transformer.named_parameters()}
        return names | {"vae." + n for n in self.vae._shapes}

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        """Names prefixed `transformer.` / `vae.` are routed to the sub-models (AutoWeightsLoader role, :752-754)."""
        tw, vw = [], []
        for n, w in weights:
            if n.startswith("transformer."):
                tw.append((n[len("transformer."):], w))
            elif n.startswith("vae."):
                vw.append((n[len("vae."):], w))
        loaded = {"transformer." + n for n in self.transformer.load_weights(tw)}
        loaded |= {"vae." + n for n in self.vae.load_weights(vw)}
        return loaded
