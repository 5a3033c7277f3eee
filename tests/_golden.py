"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


def jload(arr):
    return json.loads(bytes(arr.tobytes()).decode())


def meta_from_arrays(n, cyx, area, score):
    """dense golden arrays -> list[dict[id -> {center_yx, area, score}]]"""
    out = []
    for b in range(len(n)):
        out.append({i + 1: {'center_yx': (int(cyx[b, i, 0]), int(cyx[b, i, 1])),
                            'area': int(area[b, i]),
                            'score': float(score[b, i])}
                    for i in range(int(n[b]))})
    return out


def ids_from_arrays(n, k, v):
    return [{int(k[b, i]): int(v[b, i]) for i in range(int(n[b]))}
            for b in range(len(n))]


def cos_emb_large_cases():
    """(name, params, inputs, golden) of tests/golden/cos_emb_large.npz: the inputs are
    regenerated from the seed and checked against the stored digest"""
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    g = load('cos_emb_large')
    for name in jload(g['names']):
        p = jload(g[f'{name}__params'])
        inp = syn.make_embedding_inputs(p['B'], p['D'], p['H'], p['W'], p['L'], seed=p['seed'],
                                        bf16=p['bf16'])
        digest = syn.input_digest(inp['embedding_pred'], inp['embedding_lut'],
                                  inp['embedding_indices'])
        assert digest == jload(g[f'{name}__digest']), f'{name}: regenerated inputs differ'
        yield name, p, inp, g


def check_cos_emb_large_grad(name, p, g, grad, rtol, atol, sums_rtol):
    """full gradient [B,D,H,W] against the golden's sampled pixel rows and the three f64
    projections of the whole tensor (sum, sum |.|, random projection)"""
    grad = np.asarray(grad, np.float64)
    n_px = p['B'] * p['H'] * p['W']
    rows = grad.transpose(0, 2, 3, 1).reshape(-1, p['D'])[g[f'{name}__grad_pixels']]
    np.testing.assert_allclose(rows, g[f'{name}__grad_rows'], rtol=rtol, atol=atol, err_msg=name)
    rng = np.random.default_rng(int(g[f'{name}__proj_seed']))       # replay the generator's draws
    pix = np.sort(rng.choice(n_px, size=min(96, n_px), replace=False))
    assert (pix == g[f'{name}__grad_pixels']).all()
    proj = rng.standard_normal(grad.shape)
    want = g[f'{name}__grad_sums']
    scale = want[1]                                  # sum |grad|: the magnitude the sums live on
    got = np.array([grad.sum(), np.abs(grad).sum(), (grad * proj).sum()])
    np.testing.assert_allclose(got, want, rtol=0, atol=sums_rtol * scale, err_msg=name)


def aten_softmax_argmax(logits):
    """a1, numpy restatement (independent of the C oracle and of the kernels) of the reference's
    `softmax(dim=1)` -> `max(dim=1)` (semantic.py:52-53) in ATen's CPU arithmetic: per pixel,
    sequentially over the classes, e = Sleef expf_u10(x - max) in its FMA form, S = fp32 running
    sum, p = e / S; first index of the largest p.  -> (index uint8, max p float32)"""
    import numpy as np
    f32 = np.float32
    ld = np.longdouble

    def fma(a, b, c):                    # exact product, one rounding (64-bit significand in between)
        return (np.asarray(a, ld) * np.asarray(b, ld) + np.asarray(c, ld)).astype(f32)

    x = np.asarray(logits, f32)
    d0 = (x - x.max(axis=1, keepdims=True)).astype(f32)
    d = np.maximum(d0, f32(-104.0))                  # (below Sleef's cut-off the result is 0: see the end)
    q = np.rint((d * f32(1.442695040888963407359924681001892137426645954152985934135449406931)).astype(f32))
    s = fma(q, f32(-0.693145751953125), d)
    s = fma(q, f32(-1.428606765330187045e-06), s)
    u = np.full_like(s, f32(0.000198527617612853646278381))
    for c in (0.00139304355252534151077271, 0.00833336077630519866943359, 0.0416664853692054748535156,
              0.166666671633720397949219, 0.5):
        u = fma(u, s, f32(c))
    u = (fma((s * s).astype(f32), u, s) + f32(1.0)).astype(f32)
    qi = q.astype(np.int32)
    e = (np.ldexp(u, qi >> 1).astype(f32) * np.ldexp(f32(1.0), qi - (qi >> 1)).astype(f32)).astype(f32)
    e = np.where(d0 >= f32(-104.0), e, f32(0.0)).astype(f32)
    S = np.zeros_like(e[:, 0])
    for c in range(e.shape[1]):
        S = (S + e[:, c]).astype(f32)
    p = (e / S[:, None]).astype(f32)
    return p.argmax(axis=1).astype(np.uint8), p.max(axis=1)
