"""CPU model of the index algebra of vae.hip's conv_bordered_kernel (the launcher's run / tile arithmetic, the DMA piece ->
LDS row mapping with its XOR swizzle, the row-shifted A blocks of the (ky, channel chunk) K-tiles, the kx taps as row offsets,
the interior store and the zero-border duty of every run).  The HIP kernel is tested on the GPU (tests/test_gpu_vae_ops.py);
this file pins the arithmetic it shares with the launcher on shapes the GPU tests do not enumerate: ragged rows, several runs
per tile, runs shorter and longer than a row, 1x1.  Reference: F.pad + conv (autoencoder_kl_qwenimage.py:69-84)."""
import itertools

import numpy as np
import pytest
import torch

CONV_MAX_SEG = 4


def run_len(win, mt, gran):
    L = (win + gran - 1) // gran * gran
    L = min(L, mt)
    while mt % L or mt // L > CONV_MAX_SEG:
        L += gran
    return L


def model_conv(xb, w, bias, res_b, mt, nt, gran, kc):
    """xb [Hp, Wp, Cin] zero-bordered, w [Cout, ks, ks, Cin]; returns the bordered output the kernel would write (NaN where
    it writes nothing) and the number of writes per output pixel."""
    Hp, Wp, Cin = xb.shape
    H, W = Hp - 2, Wp - 2
    Cout, nk = w.shape[0], w.shape[1]
    flat = xb.reshape(Hp * Wp, Cin)
    npix = Hp * Wp
    L = run_len(W, mt, gran)
    seg, rpr = mt // L, (W + L - 1) // L
    nruns = H * rpr
    ntiles = (nruns + seg - 1) // seg
    rpp = 1024 // (kc * 2)                                   # rows per DMA piece
    y = np.full((npix, Cout), np.nan, np.float64)
    writes = np.zeros(npix, np.int64)

    def rows(first, count):                                  # a DMA block: rows outside the image are zeros (descriptor range)
        out = np.zeros((count, Cin))
        idx = np.arange(first, first + count)
        ok = (idx >= 0) & (idx < npix)
        out[ok] = flat[idx[ok]]
        return out

    for tile in range(ntiles):
        for n0 in range(0, Cout, nt):
            cols = slice(n0, min(n0 + nt, Cout))
            for r in range(seg):
                run = min(tile * seg + r, nruns - 1)         # runs past the image repeat the last one (computed, not stored)
                yy = run // rpr
                origin = (yy + 1) * Wp + (run - yy * rpr) * L    # raster index of the pixel LEFT of the run
                acc = np.zeros((L, cols.stop - cols.start))
                for ky in range(nk):
                    shift = (ky - 1) * Wp if nk == 3 else 0
                    for cc in range(Cin // kc):
                        blk = rows(origin + shift, L + rpp)[:, cc * kc:(cc + 1) * kc]     # ONE block for the nk taps of this row
                        for kx in range(nk):
                            dx = kx if nk == 3 else 1
                            acc += blk[dx:dx + L] @ w[cols, ky, kx, cc * kc:(cc + 1) * kc].T
                real = tile * seg + r
                if real >= nruns:
                    continue
                x0 = (real - (real // rpr) * rpr) * L
                ln = min(L, W - x0)
                m = (yy + 1) * Wp + x0 + 1
                v = acc[:ln] + bias[cols]
                if res_b is not None:
                    v = v + res_b.reshape(npix, Cout)[m:m + ln, cols]
                y[m:m + ln, cols] = v
                if n0 == 0:
                    writes[m:m + ln] += 1
                first, last = x0 == 0, x0 + ln == W

                def zero(m_first, count):
                    y[m_first:m_first + count, cols] = 0.0
                    if n0 == 0:
                        writes[m_first:m_first + count] += 1

                if first:
                    zero((yy + 1) * Wp, 1)
                if last:
                    zero((yy + 1) * Wp + W + 1, 1)
                if yy == 0:
                    zero(x0 + 1 - int(first), ln + int(first) + int(last))
                if yy == H - 1:
                    zero((Hp - 1) * Wp + x0 + 1 - int(first), ln + int(first) + int(last))
    return y.reshape(Hp, Wp, Cout), writes.reshape(Hp, Wp)


@pytest.mark.parametrize("H,W,cin,cout,ks,cfg", [
    (5, 70, 32, 16, 3, (256, 96, 64, 32)),       # two runs of 64 per row + ragged, four runs per tile
    (7, 20, 32, 24, 3, (256, 96, 64, 32)),       # rows shorter than a run
    (3, 300, 32, 8, 3, (256, 96, 64, 32)),       # rows longer than a tile
    (9, 130, 64, 40, 3, (512, 192, 128, 16)),    # 128-pixel waves, 16-channel K-tiles, ragged second run
    (6, 256, 32, 16, 3, (512, 192, 128, 16)),    # two whole rows per tile
    (4, 33, 64, 200, 1, (256, 96, 64, 32)),      # 1x1, several channel tiles with a ragged last one
])
def test_bordered_conv_index_model_matches_conv2d(H, W, cin, cout, ks, cfg):
    mt, nt, gran, kc = cfg
    g = np.random.default_rng(H * 1000 + W)
    x = g.standard_normal((H, W, cin))
    w = g.standard_normal((cout, ks, ks, cin)) * 0.1
    b = g.standard_normal(cout)
    res = g.standard_normal((H, W, cout))
    xb = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    rb = np.pad(res, ((1, 1), (1, 1), (0, 0)))
    y, writes = model_conv(xb, w, b, rb, mt, nt, gran, kc)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(w).permute(0, 3, 1, 2),
                                     torch.from_numpy(b), padding=ks // 2)[0].permute(1, 2, 0).numpy() + res
    assert np.isfinite(y).all()                                   # every pixel of the bordered raster is written ...
    assert (writes[1:-1, 1:-1] == 1).all() and (writes >= 1).all()    # ... interior exactly once, border at least once
    assert np.abs(y[1:-1, 1:-1] - ref).max() <= 1e-9
    assert np.abs(y[0]).max() == 0 and np.abs(y[-1]).max() == 0 and np.abs(y[:, 0]).max() == 0 and np.abs(y[:, -1]).max() == 0


@pytest.mark.parametrize("rb", [64, 32])
def test_lds_swizzle_is_a_bijection_and_conflict_free_at_any_row_offset(rb):
    """DMA side: lane L of piece p writes LDS bytes p*1024 + 16 L with the logical chunk (L % CPR) ^ swz(row); fragment side:
    row r, logical chunk c is read at r*RB + ((c ^ swz(r)) << 4).  The two must agree, and 16 consecutive rows (any start: the
    kx taps shift the rows by 0 / 1 / 2) must hit 16 different 16-byte bank groups of the 256-byte LDS line."""
    cpr = rb // 16
    swz = (lambda r: (r >> 2) & 3) if rb == 64 else (lambda r: (r >> 3) & 1)
    rows = 1024 // rb * 5                                    # five pieces
    lds = {}
    for p, lane in itertools.product(range(5), range(64)):
        row = p * (1024 // rb) + lane // cpr
        logical = (lane % cpr) ^ swz(lane // cpr)            # what the kernel computes from the lane alone
        assert swz(row) == swz(lane // cpr)                  # (pieces start at multiples of 16 / 32 rows)
        lds[p * 1024 + 16 * lane] = (row, logical)
    for r, c in itertools.product(range(rows), range(cpr)):
        assert lds[r * rb + ((c ^ swz(r)) << 4)] == (r, c)
    for start, c in itertools.product(range(rows - 16), range(cpr)):
        groups = {((r * rb + ((c ^ swz(r)) << 4)) % 256) // 16 for r in range(start, start + 16)}
        assert len(groups) == 16
