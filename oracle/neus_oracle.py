"""TEST INFRASTRUCTURE ONLY - numpy restatement of the NeuS render arithmetic (python/jnerf/models/samplers/neus_render/renderer.py), the checker for
jnerf_amd/neus_renderer.py and the HIP compositing kernel (csrc/neus.hip).  Only tests/ may import this.

PINNED THROUGH A STAND-IN (since round 3): the reference's NeuS is Jittor Python and Jittor is neither in /root/reference nor installable, so the reference cannot
run as it is.  Its renderer / network FILES are however executed, unmodified and from where they lie, over oracle/jt_shim (torch primitives with Jittor's semantics
restated where the libraries differ) by tests/golden/make_golden_pyref.py, and tests/test_pyref_golden.py holds these functions (sample_pdf_det, composite) and
jnerf_amd's renderer to the resulting vectors (tests/golden/golden_pyref_v1.npz; agreement: one fp32 ulp).  That checks every formula, index and ordering decision of the
reference's source; what it cannot check is Jittor's own primitives (norm's eps, safe_clip's gradient, cumprod as exp-sum-log), which are restated from Jittor's
published source in the stand-in's header - for those, parity remains unpinned.  Each function still cites the reference lines it follows.  Loops are explicit (per ray,
per sample) so that nothing is shared with the vectorised torch / HIP implementations under test.
"""
import numpy as np


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def sample_pdf_det(bins, weights, n_samples):
    """renderer.py:41-66 with det=True: inverse-CDF sampling at the n_samples bin centres of [0,1]"""
    bins, weights = np.asarray(bins, np.float64), np.asarray(weights, np.float64)
    out = np.zeros((bins.shape[0], n_samples))
    for r in range(bins.shape[0]):
        w = weights[r] + 1e-5                                            # :43
        pdf = w / w.sum()                                                # :44
        cdf = np.concatenate([[0.0], np.cumsum(pdf)])                    # :45-46
        for j in range(n_samples):
            u = 0.5 / n_samples + j * (1.0 - 1.0 / n_samples) / max(n_samples - 1, 1)      # linspace(0.5/n, 1-0.5/n, n) :49
            ind = int(np.searchsorted(cdf, u, side="right"))             # :55
            below, above = max(ind - 1, 0), min(ind, len(cdf) - 1)       # :56-57
            denom = cdf[above] - cdf[below]                              # :64
            if denom < 1e-5:
                denom = 1.0                                              # :65
            t = (u - cdf[below]) / denom                                 # :66
            out[r, j] = bins[r, below] + t * (bins[r, above] - bins[r, below])
    return out


def neus_alpha(sdf, true_cos, dist, inv_s, cos_anneal_ratio):
    """renderer.py:216-236 for one section: (alpha clipped to [0,1], p, c)"""
    relu = lambda v: max(v, 0.0)
    iter_cos = -(relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + relu(-true_cos) * cos_anneal_ratio)     # :218-219
    next_sdf = sdf + iter_cos * dist * 0.5                               # :222
    prev_sdf = sdf - iter_cos * dist * 0.5                               # :223
    prev_cdf, next_cdf = sigmoid(prev_sdf * inv_s), sigmoid(next_sdf * inv_s)
    p, c = prev_cdf - next_cdf, prev_cdf                                 # :228-229
    return min(max((p + 1e-5) / (c + 1e-5), 0.0), 1.0), p, c             # :231


def composite(sdf, true_cos, dists, inv_s, color, inside, bg_alpha, bg_color, cos_anneal_ratio):
    """renderer.py:216-252: per ray the section opacities, the blend with the background model outside the unit sphere, transmittance weights and the colour.
    sdf / true_cos / dists / inside [B, n]; color [B, n, 3]; bg_alpha [B, n + n_out] or None; bg_color [B, n + n_out, 3].  Returns (color [B,3], weights, alpha)."""
    B, n = sdf.shape
    total = n if bg_alpha is None else bg_alpha.shape[1]
    out_c, out_w, out_a = np.zeros((B, 3)), np.zeros((B, total)), np.zeros((B, total))
    for r in range(B):
        T = 1.0
        for i in range(total):
            if i < n:
                a, _, _ = neus_alpha(float(sdf[r, i]), float(true_cos[r, i]), float(dists[r, i]), float(inv_s), cos_anneal_ratio)
                col = np.asarray(color[r, i], np.float64)
                if bg_alpha is not None:                                 # :238-244
                    a = a * inside[r, i] + bg_alpha[r, i] * (1.0 - inside[r, i])
                    col = col * inside[r, i] + bg_color[r, i] * (1.0 - inside[r, i])
            else:
                a, col = float(bg_alpha[r, i]), np.asarray(bg_color[r, i], np.float64)
            w = a * T                                                    # :248
            T *= 1.0 - a + 1e-6
            out_a[r, i], out_w[r, i] = a, w
            out_c[r] += w * col                                          # :251
    return out_c, out_w, out_a


def up_sample_weights(z_vals, sdf, radius, inv_s):
    """renderer.py:117-162 up to the weights handed to sample_pdf: radius [B, n] = |o + d z|"""
    B, n = z_vals.shape
    w = np.zeros((B, n - 1))
    for r in range(B):
        T, prev_cos = 1.0, 0.0
        for i in range(n - 1):
            inside = (radius[r, i] < 1.0) or (radius[r, i + 1] < 1.0)   # :124
            mid = (sdf[r, i] + sdf[r, i + 1]) * 0.5                      # :128
            cos = (sdf[r, i + 1] - sdf[r, i]) / (z_vals[r, i + 1] - z_vals[r, i] + 1e-5)      # :129
            c = min(prev_cos, cos)                                       # :147-149
            prev_cos = cos
            c = min(max(c, -1e3), 0.0) * float(inside)                   # :150
            dist = z_vals[r, i + 1] - z_vals[r, i]
            prev_cdf = sigmoid((mid - c * dist * 0.5) * inv_s)
            next_cdf = sigmoid((mid + c * dist * 0.5) * inv_s)
            a = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)         # :158
            w[r, i] = a * T
            T *= 1.0 - a + 1e-6
    return w
