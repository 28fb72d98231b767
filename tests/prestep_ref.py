"""TEST INFRASTRUCTURE: numpy restatement of the reference's scene set-up loops (the "pre-step" of
the path: SURVEY.md 8f rank 3), float32 operation by operation, so that the host library's loops and
the GPU pre-step kernels can be held to something that is neither of them.

  mip_chain      ImageTexture ctor + col_at_uv_mipmap   reference src/image_texture.cpp:60-158
  handle_wrapping                                        reference include/texture/texture_common.h:22-53
  env_cdfs       ArraySampling2D / ArraySampling1D      reference include/rng/sampling.h:107-197
"""
import numpy as np

F = np.float32
WRAP_CLAMP, WRAP_MIRROR, WRAP_REPEAT = 0, 1, 2


def handle_wrapping(coord, mode):
    coord = coord.astype(F)
    if mode == WRAP_CLAMP:
        return np.minimum(np.maximum(coord, F(0)), F(1))          # std::clamp(coord, 0.f, 1.f)
    ipart = np.trunc(coord).astype(np.int32)                      # static_cast<int>(coord)
    frac = (coord - ipart.astype(F)).astype(F)
    neg = np.signbit(frac)
    if mode == WRAP_REPEAT:
        return np.where(neg, F(1) + frac, frac).astype(F)
    odd = (ipart % 2) != 0                                        # int_part % 2 (true for -1 and 1)
    return np.where(neg, np.where(odd, np.abs(frac), F(1) + frac), frac).astype(F)


def _mix(x, y, a):   # glm::mix(x, y, a) = x * (1 - a) + y * a
    a = a[..., None]
    return (x * (F(1) - a) + y * a).astype(F)


def col_at_uv(level, w, h, u, v, wrap_u, wrap_v):
    """col_at_uv_mipmap on one level ([h, w, 3]) for arrays of uv."""
    pu = (handle_wrapping(u, wrap_u) * F(w)).astype(F)
    pv = (handle_wrapping(v, wrap_v) * F(h)).astype(F)
    cx = np.clip(np.trunc(pu).astype(np.int64), 0, w - 1)
    cy = np.clip(np.trunc(pv).astype(np.int64), 0, h - 1)
    nx = np.clip(cx + 1, 0, w - 1)
    ny = np.clip(cy + 1, 0, h - 1)
    fx = (pu - cx.astype(F)).astype(F)
    fy = (pv - cy.astype(F)).astype(F)
    a = _mix(level[cy, cx], level[cy, nx], fx)
    b = _mix(level[ny, cx], level[ny, nx], fx)
    return _mix(a, b, fy)


TAPS = [(0.37487566, -0.75777, -0.75777), (0.37487566, 0.75777, -0.75777), (0.37487566, 0.75777, 0.75777),
        (0.37487566, -0.75777, 0.75777), (-0.12487566, -2.907, 0.0), (-0.12487566, 2.907, 0.0),
        (-0.12487566, 0.0, -2.907), (-0.12487566, 0.0, 2.907)]


def mip_chain(level0, wrap_u, wrap_v):
    """All levels of the reference's chain, level 0 first (list of [h, w, 3] float32 arrays)."""
    img = np.ascontiguousarray(level0, dtype=F)
    h, w = img.shape[:2]
    num_levels = min(int(np.ceil(np.log2(F(min(w, h))))), 15)
    levels = [img]
    pw, ph = w, h
    for _ in range(1, num_levels):
        nw, nh = max(pw // 2, 1), max(ph // 2, 1)
        inv_x, inv_y = F(1) / F(pw), F(1) / F(ph)
        xs, ys = np.meshgrid(np.arange(nw), np.arange(nh))
        cu = ((F(2) * xs.astype(F)) * inv_x).astype(F)
        cv = ((F(2) * ys.astype(F)) * inv_y).astype(F)
        total = np.zeros((nh, nw, 3), dtype=F)
        for wgt, ox, oy in TAPS:
            u = (cu + F(ox) * inv_x).astype(F)
            v = (cv + F(oy) * inv_y).astype(F)
            total = (total + F(wgt) * col_at_uv(levels[-1], pw, ph, u, v, wrap_u, wrap_v)).astype(F)
        total[total < 0] = F(0)
        levels.append(total)
        pw, ph = nw, nh
    return levels


def _cdf_1d(values):
    """ArraySampling1D: running float sum, then normalisation (uniform when the integral is 0)."""
    n = len(values)
    cdf = np.zeros(n + 1, dtype=F)
    acc = F(0)
    av = np.abs(values.astype(F))
    for i in range(n):
        acc = F(acc + av[i])
        cdf[i + 1] = acc
    func_int = cdf[n]
    if func_int == 0:
        cdf = (np.arange(n + 1, dtype=F) / F(n)).astype(F)
    else:
        cdf = (cdf / func_int).astype(F)
    return cdf, func_int


def env_cdfs(img):
    """(row_cdf [h + 1], col_cdfs [h, w + 1]) of a lat-long image as ArraySampling2D builds them."""
    img = np.ascontiguousarray(img, dtype=F)
    h, w = img.shape[:2]
    v = ((np.arange(h, dtype=F) + F(0.5)) / F(h)).astype(F)
    sin_el = np.sin(np.pi * v.astype(np.float64)).astype(F)        # std::sin(double) narrowed to float
    lum = ((img[..., 0] * F(0.212671) + img[..., 1] * F(0.715160)).astype(F) + img[..., 2] * F(0.072169)).astype(F)
    lum = (lum * sin_el[:, None]).astype(F)
    cols = np.zeros((h, w + 1), dtype=F)
    ints = np.zeros(h, dtype=F)
    for y in range(h):
        cols[y], ints[y] = _cdf_1d(lum[y])
    rows, _ = _cdf_1d(ints)
    return rows, cols
