"""CPU: the error behaviour of the C-ABI (include/omni_cdna4.h "Conventions"; SURVEY.md §8b: "returning int status (0 = ok,
negative = enum) and never throwing/aborting").  Every int-returning entry point is called with ONE argument of an otherwise
valid call made invalid and must answer with the documented status — BEFORE it touches the device: the pointers below are
made-up addresses that are never dereferenced, so these calls are legal on a host without a GPU (they compute nothing).
The order of the checks is part of the contract a binder sees: null / size errors (BAD_ARG) before shape limits (UNSUPPORTED)
before alignment (ALIGN).  The Python wrappers turn any non-zero status into OmniNativeError (the reference's error path:
exceptions caught at diffusion/worker/gpu_worker.py:266-274 and stringified into DiffusionOutput.error)."""
import ctypes as C

import pytest

OK, BAD_ARG, UNSUPPORTED, LAUNCH, ALIGN = 0, -1, -2, -3, -4
P = 0x7F0000010000          # a "device pointer": 16-byte aligned, never dereferenced
P8 = P + 8                  # 8-byte aligned only
P2 = P + 2                  # bf16-aligned only


@pytest.fixture(scope="module")
def lib():
    from vllm_omni_amd import _native as N

    return N.lib()


def _call(lib, name, args):
    return getattr(lib, name)(*args)


def _mutations(lib, name, base, cases):
    """`base`: a list of positional arguments that passes validation; `cases`: (index, value, expected status)."""
    for idx, val, want in cases:
        args = list(base)
        args[idx] = val
        got = _call(lib, name, args)
        assert got == want, f"{name}: argument {idx} = {val!r}: status {got}, expected {want}"


def test_status_strings_and_identity(lib):
    from vllm_omni_amd import _native as N

    assert lib.omni_abi_version() == N.ABI_VERSION
    assert lib.omni_build_arch() == b"gfx950"
    seen = set()
    for code in (OK, BAD_ARG, UNSUPPORTED, LAUNCH, ALIGN):
        s = lib.omni_status_string(code)
        assert s and s != b"unknown status" and s not in seen
        seen.add(s)
    assert lib.omni_status_string(-99) == b"unknown status"
    with pytest.raises(N.OmniNativeError, match="status -2"):
        N.check(UNSUPPORTED, "probe")
    N.check(OK, "probe")


def test_elementwise_entry_points_reject_bad_arguments(lib):
    # omni_adaln_modulate(x, ldx, y, ldy, rows, D, scale, shift, mod_item_stride, row_item_map, rows_per_item, eps, stream)
    base = [P, 3072, P, 3072, 128, 3072, P, P, 6 * 3072, None, 128, 1e-6, None]
    _mutations(lib, "omni_adaln_modulate", base, [
        (0, None, BAD_ARG), (2, None, BAD_ARG), (6, None, BAD_ARG), (7, None, BAD_ARG), (4, 0, BAD_ARG), (5, -8, BAD_ARG),
        (10, 0, BAD_ARG),                               # no row_item_map AND no rows_per_item: rows cannot be attributed to items
        (5, 3076, UNSUPPORTED), (5, 8200, UNSUPPORTED),  # D % 8, D > 8192
        (0, P8, ALIGN), (2, P2, ALIGN), (6, P8, ALIGN), (1, 3074, ALIGN), (3, 3073, ALIGN), (8, 18434, ALIGN)])
    # ..._ex adds y_k32_rows in front of the stream
    ex = base[:12] + [128, None]
    _mutations(lib, "omni_adaln_modulate_ex", ex, [(12, -1, BAD_ARG), (12, 64, BAD_ARG), (5, 3080, UNSUPPORTED)])   # blocked rows < rows; D % 32
    # omni_adaln_modulate_fp8(x, ldx, rows, D, scale, shift, stride, map, rows_per_item, eps, y, y_k32_rows, y8, y8_rows, y8_scale, stream)
    f8 = [P, 3072, 128, 3072, P, P, 6 * 3072, None, 128, 1e-6, None, 0, P, 128, P, None]
    _mutations(lib, "omni_adaln_modulate_fp8", f8, [
        (12, None, BAD_ARG), (14, None, BAD_ARG), (13, 64, BAD_ARG), (3, 3104, UNSUPPORTED), (12, P + 4, ALIGN)])
    args = list(f8)
    args[10], args[11] = P, 0                            # the optional bf16 copy must come with its blocked row count
    assert _call(lib, "omni_adaln_modulate_fp8", args) == BAD_ARG
    # omni_adaln_modulate_pair(a, b, D, mod_item_stride, eps, stream) — ABI v12: two row groups (image / text stream) in one launch
    from vllm_omni_amd import _native as N

    def stream_rec(**kw):
        r = N.AdalnStream(x=P, y=P, rows=128, scale=P, shift=P, row_item_map=P, rows_per_item=0, y_k32_rows=0, y8=None, y8_rows=0,
                          y8_scale=None)
        for k, v in kw.items():
            setattr(r, k, v)
        return r

    def pair(a=None, b=None, D=3072, stride=6 * 3072):
        a = stream_rec() if a is None else a
        b = stream_rec(rows=64) if b is None else b
        return lib.omni_adaln_modulate_pair(C.byref(a), C.byref(b), D, stride, 1e-6, None)

    assert lib.omni_adaln_modulate_pair(None, C.byref(stream_rec()), 3072, 6 * 3072, 1e-6, None) == BAD_ARG
    assert pair(D=0) == BAD_ARG and pair(a=stream_rec(x=None)) == BAD_ARG and pair(b=stream_rec(scale=None)) == BAD_ARG
    assert pair(b=stream_rec(y=None)) == BAD_ARG                              # neither a bf16 nor an fp8 output
    assert pair(a=stream_rec(row_item_map=None)) == BAD_ARG                  # rows cannot be attributed to items
    assert pair(a=stream_rec(y_k32_rows=64)) == BAD_ARG                      # blocked rows < rows
    assert pair(b=stream_rec(y8=P8, y8_rows=128)) == BAD_ARG                 # an fp8 copy without its scales
    assert pair(b=stream_rec(y8=P8, y8_rows=128, y8_scale=P)) == BAD_ARG     # the bf16 copy beside an fp8 copy is K32-blocked
    assert pair(D=3076) == UNSUPPORTED and pair(D=8200) == UNSUPPORTED
    assert pair(a=stream_rec(y_k32_rows=128), D=3080) == UNSUPPORTED         # K32-blocked output: D % 32
    assert pair(b=stream_rec(y=None, y8=P8, y8_rows=128, y8_scale=P), D=3104) == UNSUPPORTED   # fp8: D % 64
    assert pair(a=stream_rec(x=P2)) == ALIGN and pair(b=stream_rec(shift=P8)) == ALIGN and pair(stride=6 * 3072 + 4) == ALIGN
    assert pair(a=stream_rec(x=None), b=stream_rec(x=P2)) == BAD_ARG        # BAD_ARG of either group before ALIGN of the other
    # omni_splitk_finish_adaln_pair(splitk_ws, nsplit, ws_rows, a, b, D, mod_item_stride, eps, stream) — ABI v13: the finish of a
    # deferred split-K GEMM (gated residual) + the AdaLN behind it, both streams in one launch
    def fin_rec(**kw):
        r = N.FinishAdalnStream(rows=512, ws_row0=0, bias=P, hidden=P, gate=P, scale=P, shift=P, row_item_map=P, rows_per_item=0,
                                y=P, y_k32_rows=0)
        for k, v in kw.items():
            setattr(r, k, v)
        return r

    def fin(ws=P, nsplit=6, ws_rows=640, a=None, b=None, D=3072, stride=6 * 3072):
        a = fin_rec() if a is None else a
        b = fin_rec(rows=128, ws_row0=512) if b is None else b
        return lib.omni_splitk_finish_adaln_pair(ws, nsplit, ws_rows, C.byref(a), C.byref(b), D, stride, 1e-6, None)

    assert fin(ws=None) == BAD_ARG and fin(D=0) == BAD_ARG and fin(ws_rows=0) == BAD_ARG
    assert lib.omni_splitk_finish_adaln_pair(P, 6, 640, None, C.byref(fin_rec()), 3072, 6 * 3072, 1e-6, None) == BAD_ARG
    assert fin(a=fin_rec(hidden=None)) == BAD_ARG and fin(b=fin_rec(gate=None)) == BAD_ARG and fin(a=fin_rec(y=None)) == BAD_ARG
    assert fin(a=fin_rec(row_item_map=None)) == BAD_ARG                      # rows cannot be attributed to items
    assert fin(b=fin_rec(rows=128, ws_row0=600)) == BAD_ARG                  # the group's rows end past the partials' rows
    assert fin(a=fin_rec(y_k32_rows=64)) == BAD_ARG                          # blocked rows < rows
    assert fin(nsplit=5) == UNSUPPORTED and fin(nsplit=1) == UNSUPPORTED     # the factors split-K produces: 2, 3, 4, 6, 8
    assert fin(D=3076) == UNSUPPORTED and fin(D=4104) == UNSUPPORTED
    assert fin(a=fin_rec(y_k32_rows=512), D=3080) == UNSUPPORTED             # K32-blocked output: D % 32
    assert fin(a=fin_rec(hidden=P2)) == ALIGN and fin(b=fin_rec(rows=128, ws_row0=512, bias=P8)) == ALIGN
    assert fin(ws=P8) == ALIGN and fin(stride=6 * 3072 + 4) == ALIGN
    # omni_rmsnorm(x, ldx, y, ldy, rows, D, weight, eps, stream)
    _mutations(lib, "omni_rmsnorm", [P, 3584, P, 3584, 64, 3584, P, 1e-6, None], [
        (6, None, BAD_ARG), (4, -1, BAD_ARG), (5, 3588, UNSUPPORTED), (6, P8, ALIGN), (3, 3585, ALIGN)])
    # omni_qk_norm_rope(x, ldx, rows, num_heads, w_img, w_txt, cos, sin, row_pos, txt_pos_end, eps, stream)
    _mutations(lib, "omni_qk_norm_rope", [P, 9216, 64, 24, P, P, P, P, P, 64, 1e-6, None], [
        (8, None, BAD_ARG), (3, 0, BAD_ARG), (0, P8, ALIGN), (1, 9220, ALIGN), (6, P + 4, ALIGN)])
    # omni_rope_interleaved(x, y, B, S, H, dh, cos, sin, stream)
    _mutations(lib, "omni_rope_interleaved", [P, P, 1, 64, 24, 128, P, P, None], [
        (1, None, BAD_ARG), (2, 0, BAD_ARG), (5, 120, UNSUPPORTED), (0, P8, ALIGN), (7, P2, ALIGN)])
    # omni_linear_smallbatch(x, ldx, B, W, bias, N, K, y, ldy, act_in, act_out, stream)
    _mutations(lib, "omni_linear_smallbatch", [P, 3072, 2, P, None, 18432, 3072, P, 18432, 1, 0, None], [
        (3, None, BAD_ARG), (5, 0, BAD_ARG), (2, 9, UNSUPPORTED), (6, 3076, UNSUPPORTED), (6, 8192, UNSUPPORTED), (3, P8, ALIGN)])
    # omni_timestep_sinusoid(t, B, dim, scale, out, stream)
    _mutations(lib, "omni_timestep_sinusoid", [P, 1, 256, 1000.0, P, None], [(0, None, BAD_ARG), (2, 255, BAD_ARG), (1, 0, BAD_ARG)])
    # omni_cfg_euler_step(pos, neg, latents, rows, C, scale, dt, dt_rows_per_item, stream); neg may be NULL (no CFG)
    cfg = [P, P, P, 4096, 64, 4.0, P, 4096, None]
    _mutations(lib, "omni_cfg_euler_step", cfg, [(2, None, BAD_ARG), (6, None, BAD_ARG), (3, 0, BAD_ARG), (4, 32, UNSUPPORTED),
                                                 (1, P8, ALIGN), (0, P2, ALIGN)])
    _mutations(lib, "omni_cfg_euler_step_ex", cfg[:8] + [0, None], [(0, None, BAD_ARG), (4, 128, UNSUPPORTED)])
    # omni_quantize_fp8_rows(x, ldx, x_k32_rows, rows, K, y8, y_rows, scale, stream)
    _mutations(lib, "omni_quantize_fp8_rows", [P, 3072, 0, 128, 3072, P, 128, P, None], [
        (5, None, BAD_ARG), (6, 64, BAD_ARG), (2, 64, BAD_ARG), (4, 3104, UNSUPPORTED), (4, 16448, UNSUPPORTED), (5, P + 4, ALIGN),
        (1, 3076, ALIGN)])


def test_attention_entry_points_reject_bad_arguments(lib):
    # omni_flash_attn_fwd(q, k, v, out, ldq, ldk, ldv, ldo, cu_seqlens, B, H, head_dim, max_seqlen, softmax_scale, stream)
    base = [P, P, P, P, 3072, 3072, 3072, 3072, P, 2, 24, 128, 4160, 0.088, None]
    _mutations(lib, "omni_flash_attn_fwd", base, [
        (0, None, BAD_ARG), (3, None, BAD_ARG), (8, None, BAD_ARG), (9, 0, BAD_ARG), (12, 0, BAD_ARG),
        (11, 64, UNSUPPORTED),                           # the kernels are built for head_dim 128 (get_supported_head_sizes)
        (1, P8, ALIGN), (3, P + 4, ALIGN), (4, 3076, ALIGN), (7, 3074, ALIGN)])
    _mutations(lib, "omni_flash_attn_fwd_ex", base[:14] + [0, None], [(14, -1, BAD_ARG), (11, 256, UNSUPPORTED)])
    # omni_flash_attn_fwd_ws(..., out_k32_rows, workspace, workspace_bytes, stream) — ABI v11; the workspace query is host arithmetic
    wb = lib.omni_flash_attn_workspace_bytes(2, 24)
    assert wb == 2 * 24 * 8 * 64 * 130 * 4 and lib.omni_flash_attn_workspace_bytes(0, 24) == 0
    _mutations(lib, "omni_flash_attn_fwd_ws", base[:14] + [0, P, wb, None], [
        (0, None, BAD_ARG), (9, 0, BAD_ARG), (11, 64, UNSUPPORTED), (15, P + 4, ALIGN), (2, P8, ALIGN)])
    # omni_flash_attn_general(params, stream) — ABI v11: the whole SDPA plug-in point (cross-attention, masks, causal, dh 64 / 128)
    from vllm_omni_amd import _native as N

    def general(**kw):
        a = N.AttnParams(q=P, k=P, v=P, out=P, ldq=1536, ldk=1536, ldv=1536, ldo=1536, cu_seqlens_q=P, cu_seqlens_k=P, B=2, H=24,
                         H_kv=24, head_dim=64, max_seqlen_q=1024, max_seqlen_k=77, softmax_scale=0.125, causal=0)
        for k_, v_ in kw.items():
            setattr(a, k_, v_)
        return lib.omni_flash_attn_general(C.byref(a), None)

    assert lib.omni_flash_attn_general(None, None) == BAD_ARG
    for bad in (dict(q=None), dict(out=None), dict(cu_seqlens_k=None), dict(B=0), dict(H_kv=0), dict(H_kv=5), dict(max_seqlen_q=0),
                dict(causal=2), dict(mask_type=1), dict(mask=P, mask_type=0), dict(mask=P, mask_type=4)):
        assert general(**bad) == BAD_ARG, bad
    assert general(head_dim=96) == UNSUPPORTED and general(head_dim=256) == UNSUPPORTED
    for bad in (dict(k=P8), dict(out=P + 4), dict(ldq=1540), dict(ldo=1538), dict(mask=P + 1, mask_type=2), dict(mask=P2, mask_type=3)):
        assert general(**bad) == ALIGN, bad
    # omni_vae_attention(q, k, v, out, B, tokens, C, ldq, ldk, ldv, ldo, scale, stream)
    va = [P, P, P, P, 1, 16384, 384, 1152, 1152, 1152, 384, 0.051, None]
    _mutations(lib, "omni_vae_attention", va, [
        (2, None, BAD_ARG), (5, 0, BAD_ARG), (6, 256, UNSUPPORTED), (4, 70000, UNSUPPORTED), (0, P8, ALIGN), (9, 1156, ALIGN),
        (10, 386, ALIGN)])
    big = list(va)
    big[5], big[8] = 1 << 21, 1152                       # tokens * ldk * 2 >= 4 GiB: the kernel's 32-bit offsets would wrap
    assert _call(lib, "omni_vae_attention", big) == UNSUPPORTED


def test_vae_entry_points_reject_bad_arguments(lib):
    from vllm_omni_amd import _native as N

    _mutations(lib, "omni_vae_rmsnorm_silu", [P, P, 1024, 384, P, 1, None], [
        (4, None, BAD_ARG), (2, 0, BAD_ARG), (3, 388, UNSUPPORTED), (3, 520, UNSUPPORTED), (1, P8, ALIGN)])
    _mutations(lib, "omni_softmax_rows", [P, 16384, 64, 16384, 0.05, None], [
        (0, None, BAD_ARG), (3, 0, BAD_ARG), (3, 1001, UNSUPPORTED), (0, P8, ALIGN), (1, 16388, ALIGN)])
    _mutations(lib, "omni_vae_upsample2x_bordered", [P, P, 1, 64, 64, 384, None], [
        (1, None, BAD_ARG), (3, 0, BAD_ARG), (5, 100, UNSUPPORTED), (0, P2, ALIGN)])

    def conv(**kw):
        p = N.ConvParams(x=P, w=P, y=P, B=1, Hin=64, Win=64, Cin=96, Cout=96, ksize=3)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    for kw, want in [(dict(x=None), BAD_ARG), (dict(w=None), BAD_ARG), (dict(B=0), BAD_ARG), (dict(Cout=0), BAD_ARG),
                     (dict(y=None), BAD_ARG),                                       # plain NHWC output is never optional
                     (dict(norm_gamma=P), BAD_ARG),                                 # the norm output needs its buffer
                     (dict(ksize=5), UNSUPPORTED), (dict(Cin=100), UNSUPPORTED), (dict(gamma=P), UNSUPPORTED),
                     (dict(downsample2x=1, upsample2x=1), UNSUPPORTED), (dict(downsample2x=1, Hin=63), UNSUPPORTED),
                     (dict(downsample2x=1, ksize=1), UNSUPPORTED), (dict(x=P8), ALIGN), (dict(w=P2), ALIGN)]:
        p = conv(**kw)
        got = lib.omni_vae_conv2d(C.byref(p), None)
        assert got == want, (kw, got, want)
    assert lib.omni_vae_conv2d(None, None) == BAD_ARG
    # omni_vae_conv2d_fuses_norm is pure host logic (no launch): only bordered rasters whose workgroup holds all channels of a pixel
    assert lib.omni_vae_conv2d_fuses_norm(None) == 0
    assert lib.omni_vae_conv2d_fuses_norm(C.byref(conv(norm_gamma=P, y_norm=P))) == 0              # plain NHWC: separate pass
    bordered = dict(x_padded=1, y_padded=1, norm_gamma=P, y_norm=P, Hin=1024, Win=1024)
    assert lib.omni_vae_conv2d_fuses_norm(C.byref(conv(**bordered))) == 1                           # 96 channels: one wave tile
    assert lib.omni_vae_conv2d_fuses_norm(C.byref(conv(Cin=384, Cout=384, **bordered))) == 0        # 384: two workgroups per pixel


def test_gemm_entry_point_rejects_bad_arguments(lib):
    from vllm_omni_amd import _native as N

    EPI_BIAS, EPI_GELU, EPI_GATE, EPI_SPLIT3, EPI_SPLIT3_QK = 0, 1, 2, 3, 4

    def params(epi=EPI_BIAS, N_=3072, K=3072, ngroups=1, **g0):
        p = N.GemmParams(ngroups=ngroups, N=N_, K=K, epilogue=epi)
        g = dict(A=P, lda=K, M=256, W=P, out=P, ldo=N_)
        g.update(g0)
        for gi in range(min(max(ngroups, 1), 2)):
            for k, v in g.items():
                setattr(p.g[gi], k, v)
        return p

    def status(p):
        return lib.omni_gemm_bf16(C.byref(p), None)

    assert lib.omni_gemm_bf16(None, None) == BAD_ARG
    for p, want in [
        (params(ngroups=0), BAD_ARG), (params(ngroups=3), BAD_ARG), (params(N_=0), BAD_ARG), (params(K=-64), BAD_ARG),
        (params(K=3080), UNSUPPORTED), (params(N_=3076), UNSUPPORTED),             # K % 32 (the ring kernel's stage), N % 8
        (params(A=None), BAD_ARG), (params(W=None), BAD_ARG), (params(out=None), BAD_ARG), (params(M=0), BAD_ARG),
        (params(A=P8), ALIGN), (params(W=P2), ALIGN), (params(lda=3076), ALIGN), (params(out=P + 4), ALIGN), (params(ldo=3074), ALIGN),
        (params(epi=EPI_GATE), BAD_ARG),                                           # gate-residual without res / gate
        (params(epi=EPI_GATE, res=P, gate=P), BAD_ARG),                            # ... without a row -> item attribution
        (params(epi=EPI_GATE, res=P, gate=P, rows_per_item=128, ldres=3074), ALIGN),
        (params(epi=EPI_SPLIT3, N_=9216), BAD_ARG),                                # split-to-q/k/v without out1 / out2
        (params(epi=EPI_SPLIT3_QK, N_=9216, out1=P, out2=P), BAD_ARG),             # fused q/k norm + RoPE without its tables
        (params(epi=7), BAD_ARG),
    ]:
        assert status(p) == want, (p.epilogue, p.N, p.K, status(p), want)
    # omni_gemm_splitk_factor(params) — ABI v13: host arithmetic (tile counts, K, workspace size); 0 = not a valid call
    assert lib.omni_gemm_splitk_factor(None) == 0 and lib.omni_gemm_splitk_factor(C.byref(params(A=None))) == 0
    assert lib.omni_gemm_splitk_factor(C.byref(params())) == 1                    # no workspace: never split

    def small(N_, K, epi=EPI_GATE):                                                # one 256x256 CFG pair: 512 image + 128 text rows
        p = params(epi=epi, N_=N_, K=K, ngroups=2, res=P, gate=P, rows_per_item=64, ldres=N_, out1=P, out2=P)
        p.g[0].M, p.g[1].M = 512, 128
        p.splitk_ws, p.splitk_ws_floats = P, 8 * 640 * N_
        return p

    assert lib.omni_gemm_splitk_factor(C.byref(small(3072, 3072))) == 6           # 36 tiles x 6 = 216 workgroups, 8 K-tiles each
    assert lib.omni_gemm_splitk_factor(C.byref(small(3072, 12288))) == 6
    q = small(9216, 3072, epi=EPI_SPLIT3)
    q.split_n = 3072
    assert lib.omni_gemm_splitk_factor(C.byref(q)) == 2                            # 108 tiles x 2
    big = small(3072, 3072)
    big.g[0].M = 8192                                                              # 33 row tiles: fills the chip without a split
    assert lib.omni_gemm_splitk_factor(C.byref(big)) == 1
    big.kernel_hint = 5                                                            # OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH on a call that
    assert status(big) == UNSUPPORTED                                              # does not split: refused, never another path
    p = params(epi=EPI_SPLIT3, N_=9216, out1=P, out2=P)
    p.split_n = 3000                                                               # N != 3 * split_n, split_n % 32
    assert status(p) == UNSUPPORTED
    p = params(a_k32_rows=128)                                                     # blocked A with fewer rows than M
    assert status(p) == BAD_ARG
    p = params(epi=EPI_GATE, res=P, gate=P, rows_per_item=128, ldres=3072, out_k32_rows=256)
    assert status(p) == UNSUPPORTED                                                # K32-blocked output: bias / GELU epilogues only
    p = params()
    p.w_k32_blocked = 2
    assert status(p) == BAD_ARG
    p = params()
    p.fp8 = 2
    assert status(p) == BAD_ARG
    p = params()
    p.fp8 = 1                                                                      # fp8 needs the blocked layouts and both scale vectors
    assert status(p) == UNSUPPORTED
    p.w_k32_blocked = 1
    assert status(p) == BAD_ARG
    p.g[0].a_scale, p.g[0].w_scale = P, P
    assert status(p) == UNSUPPORTED                                                # ... and K64-blocked activations
    p.g[0].a_k32_rows = 256
    p.g[0].w_scale = P + 4
    assert status(p) == ALIGN


def test_dit_entry_points_reject_bad_arguments(lib):
    from vllm_omni_amd import _native as N

    layers = (N.DitLayerWeights * 2)()
    w = N.DitWeights(num_layers=2, num_heads=24, head_dim=128, joint_dim=3584, in_channels=64, out_channels_packed=64,
                     layers=C.cast(layers, C.POINTER(N.DitLayerWeights)))
    need = lib.omni_dit_workspace_bytes(C.byref(w), 4096, 64, 1)
    assert need > 4096 * 3072 * 2 * 4                                # at least hidden + normed + q/k/v of the image stream
    assert lib.omni_dit_workspace_bytes(C.byref(w), 8192, 128, 2) > need                 # grows with the batch
    assert lib.omni_dit_workspace_bytes(None, 4096, 64, 1) == 0
    assert lib.omni_dit_workspace_bytes(C.byref(w), 0, 64, 1) == 0
    assert lib.omni_dit_workspace_bytes(C.byref(w), 4096, 64, 0) == 0

    def batch(**kw):
        b = N.DitBatch(n_items=1, n_img_rows=4096, n_txt_rows=64, n_joint_rows=4160, n_temb=1, max_seqlen=4160, workspace=P,
                       workspace_bytes=need)
        for k, v in kw.items():
            setattr(b, k, v)
        return b

    fwd = lambda ww, bb: lib.omni_dit_forward(C.byref(ww) if ww is not None else None, C.byref(bb) if bb is not None else None, None)  # noqa: E731
    assert fwd(None, batch()) == BAD_ARG and fwd(w, None) == BAD_ARG
    assert fwd(w, batch(workspace=None)) == BAD_ARG
    assert fwd(w, batch(n_joint_rows=4000)) == BAD_ARG                # joint rows != image + text rows
    assert fwd(w, batch(n_img_rows=0, n_joint_rows=64)) == BAD_ARG
    assert fwd(w, batch(workspace_bytes=need - 1)) == BAD_ARG         # the caller-owned workspace is too small: refused, not overrun
    nol = N.DitWeights(num_layers=2, num_heads=24, head_dim=128, joint_dim=3584, in_channels=64, out_channels_packed=64)
    assert fwd(nol, batch()) == BAD_ARG                               # no layer table
    w64 = N.DitWeights(num_layers=2, num_heads=24, head_dim=64, joint_dim=3584, in_channels=64, out_channels_packed=64,
                       layers=C.cast(layers, C.POINTER(N.DitLayerWeights)))
    assert fwd(w64, batch()) == UNSUPPORTED

    b = batch()
    for name, extra in (("omni_dit_block", []), ("omni_dit_block_post", [P])):
        f = getattr(lib, name)
        ok = [C.byref(w), 0, C.byref(b), P, P, P] + extra + [None]
        for idx, val in ((1, -1), (1, 2), (3, None), (5, None)):      # layer out of range, null streams / conditioning
            args = list(ok)
            args[idx] = val
            assert f(*args) == BAD_ARG, (name, idx, val)
        small = batch(workspace_bytes=1024)
        args = list(ok)
        args[2] = C.byref(small)
        assert f(*args) == BAD_ARG, name
    assert lib.omni_dit_block_post(C.byref(w), 0, C.byref(b), P, P, P, P8, None) == ALIGN
    q, k, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.omni_dit_block_qkv(C.byref(w), 0, C.byref(b), P, P, P, None, C.byref(k), C.byref(v), None) == BAD_ARG
    assert lib.omni_dit_block_qkv(C.byref(w), 5, C.byref(b), P, P, P, C.byref(q), C.byref(k), C.byref(v), None) == BAD_ARG
    assert q.value is None                                            # nothing handed out on a refused call

    # modulation table: workspace query is host arithmetic; the call refuses a short workspace
    tb = lib.omni_dit_modulation_table_workspace_bytes(C.byref(w), 20)
    assert tb >= 20 * 3072 * 2 + 8 * 20 * 6 * 3072 * 4 and lib.omni_dit_modulation_table_workspace_bytes(C.byref(w), 0) == 0
    mt = lambda **kw: lib.omni_dit_modulation_table(*[kw.get(k, d) for k, d in (("w", C.byref(w)), ("temb", P), ("M", 20), ("table", P),  # noqa: E731
                                                                                ("ws", P), ("bytes", tb), ("stream", None))])
    assert mt(temb=None) == BAD_ARG and mt(table=None) == BAD_ARG and mt(ws=None) == BAD_ARG and mt(M=0) == BAD_ARG
    assert mt(bytes=tb - 1) == BAD_ARG
    odd = N.DitWeights(num_layers=2, num_heads=3, head_dim=40, joint_dim=3584, in_channels=64, out_channels_packed=64,
                       layers=C.cast(layers, C.POINTER(N.DitLayerWeights)))
    assert lib.omni_dit_modulation_table(C.byref(odd), P, 20, P, P, 1 << 30, None) == UNSUPPORTED


def test_every_status_returning_entry_point_is_covered():
    """The table above must grow with the header: every exported function that returns an omni_status appears in this file."""
    import os
    import re

    from vllm_omni_amd import _native as N

    src = open(os.path.abspath(__file__)).read()
    status_fns = [n for n, (res, _a) in N.PROTOTYPES.items() if res is C.c_int and n not in ("omni_abi_version",)]
    missing = [n for n in status_fns if not re.search(rf"\b{n}\b", src)]
    assert not missing, missing
