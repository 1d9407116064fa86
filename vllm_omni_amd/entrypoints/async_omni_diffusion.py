"""AsyncOmniDiffusion — the asynchronous entry point of vllm_omni/entrypoints/async_omni_diffusion.py:30-270, the class the
reference's OpenAI-compatible server binds `/v1/images/generations` to (entrypoints/openai/api_server.py:544-680): same
constructor (`model`, optional `od_config`, config kwargs), `await generate(prompt, request_id=..., num_inference_steps=...,
guidance_scale=..., height=..., width=..., negative_prompt=..., num_outputs_per_prompt=..., seed=..., **kw)` ->
OmniRequestOutput, `generate_stream`, `close` / `shutdown`, `is_running` / `is_stopped`.

What differs underneath: the reference runs the blocking `engine.step([request])` of each call on a
one-thread executor, so concurrent requests are served one after another.  Here a call SUBMITS its request to the dispatching
engine (least-loaded data-parallel group) and awaits a future that one reaper thread resolves from the workers' result queue:
requests that are in flight together are step-batched by the worker's continuous batcher (one ragged DiT forward per
denoising step for all of them), and a failing request only fails its own future."""
from __future__ import annotations

import asyncio
import threading
import uuid
from collections.abc import AsyncGenerator
from dataclasses import fields
from typing import Any

from ..diffusion.data import OmniDiffusionConfig
from ..diffusion.diffusion_engine import DiffusionEngine
from ..diffusion.request import OmniDiffusionRequest
from ..outputs import OmniRequestOutput


class AsyncOmniDiffusion:
    def __init__(self, model: str | None = None, od_config: OmniDiffusionConfig | None = None, pipeline_factory=None, **kwargs: Any):
        self.model = model
        if od_config is None:
            names = {f.name for f in fields(OmniDiffusionConfig)}
            od_config = OmniDiffusionConfig(model=model, **{k: v for k, v in kwargs.items() if k in names and k != "model"})
        elif isinstance(od_config, dict):
            od_config = OmniDiffusionConfig(**od_config)
        self.od_config = od_config
        self.engine = DiffusionEngine.make_engine(od_config, pipeline_factory=pipeline_factory)
        self._lock = threading.Lock()                 # dispatcher bookkeeping: submit() on the loop thread, _handle() on the reaper
        self._waiting: dict[int, tuple[asyncio.AbstractEventLoop, asyncio.Future, OmniDiffusionRequest]] = {}
        self._closed = False
        self._reaper = threading.Thread(target=self._reap, name="AsyncOmniDiffusion-reaper", daemon=True)
        self._reaper.start()

    # ------------------------------------------------------------------ request construction (reference :85-115)
    def _prepare_request(self, prompt: str, request_id: str | None = None, **kwargs: Any) -> OmniDiffusionRequest:
        if request_id is None:
            request_id = f"diff-{uuid.uuid4().hex[:16]}"
        names = {f.name for f in fields(OmniDiffusionRequest)}
        init = {"prompt": prompt, "request_id": request_id}
        init.update({k: v for k, v in kwargs.items() if k in names})
        return OmniDiffusionRequest(**init)

    # ------------------------------------------------------------------ result side
    def _reap(self) -> None:
        while not self._closed:
            m = self.engine._recv(0.1)
            dead = not all(p.is_alive() for p in self.engine._processes)
            with self._lock:
                if m is not None:
                    self.engine._handle(m)
                for rid in [r for r in self._waiting if r in self.engine._results]:
                    loop, fut, req = self._waiting.pop(rid)
                    out = self.engine._results.pop(rid)
                    loop.call_soon_threadsafe(self._resolve, fut, req, out)
                if dead and m is None:
                    for rid, (loop, fut, _req) in list(self._waiting.items()):
                        loop.call_soon_threadsafe(self._fail, fut, RuntimeError("a diffusion worker died"))
                    self._waiting.clear()

    def _resolve(self, fut: asyncio.Future, req: OmniDiffusionRequest, out) -> None:
        if fut.done():
            return
        try:
            fut.set_result(self.engine.to_request_output(req, out))
        except Exception as e:  # noqa: BLE001 — the request's own error
            fut.set_exception(RuntimeError(f"Diffusion generation failed: {e}"))

    @staticmethod
    def _fail(fut: asyncio.Future, exc: Exception) -> None:
        if not fut.done():
            fut.set_exception(exc)

    # ------------------------------------------------------------------ the reference's async surface (:117-229)
    async def generate(self, prompt: str, request_id: str | None = None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                       height: int | None = None, width: int | None = None, negative_prompt: str | None = None,
                       num_outputs_per_prompt: int = 1, seed: int | None = None, **kwargs: Any) -> OmniRequestOutput:
        if self._closed:
            raise RuntimeError("AsyncOmniDiffusion is closed")
        if request_id is None:
            request_id = f"diff-{uuid.uuid4().hex[:16]}"
        request = self._prepare_request(prompt=prompt, request_id=request_id, num_inference_steps=num_inference_steps,
                                        guidance_scale=guidance_scale, height=height, width=width, negative_prompt=negative_prompt,
                                        num_outputs_per_prompt=num_outputs_per_prompt, seed=seed, **kwargs)
        if self.engine.pre_process_func is not None:
            request = self.engine.pre_process_func([request])[0]
        loop = asyncio.get_running_loop()
        fut: asyncio.Future = loop.create_future()
        with self._lock:
            rid = self.engine.submit(request)
            self._waiting[rid] = (loop, fut, request)
        result = await fut
        if not result.request_id:
            result.request_id = request_id
        result.metrics.setdefault("num_inference_steps", num_inference_steps)
        result.metrics.setdefault("guidance_scale", guidance_scale)
        return result

    async def generate_stream(self, prompt: str, request_id: str | None = None, **kwargs: Any) -> AsyncGenerator[OmniRequestOutput, None]:
        """Diffusion has no token stream: one result when the image is done (reference :208-229)."""
        yield await self.generate(prompt=prompt, request_id=request_id, **kwargs)

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        self._reaper.join(timeout=2.0)
        with self._lock:
            for _rid, (loop, fut, _req) in list(self._waiting.items()):
                loop.call_soon_threadsafe(self._fail, fut, RuntimeError("AsyncOmniDiffusion was closed"))
            self._waiting.clear()
        try:
            self.engine.close()
        except Exception:  # noqa: BLE001
            pass

    def shutdown(self) -> None:
        self.close()

    def __del__(self) -> None:  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    @property
    def is_running(self) -> bool:
        return not self._closed

    @property
    def is_stopped(self) -> bool:
        return self._closed
