"""Soundness of the bundle culling (trace.hip.hpp) on the CPU model (tests/cull_model.py): a triangle the culling drops
is never accepted by the exact per-ray test for ANY ray of the bundle -- hammered with random and adversarial cases
(rays aimed exactly at vertices and edges, degenerate and tiny triangles, wide and narrow bundles, noisy apexes)."""
import numpy as np
import pytest

import cull_model as cm

f32 = np.float32


def _bundle(rng, kind):
    C = (rng.normal(size=3) * rng.choice([1.0, 10.0, 60.0])).astype(f32)
    dc = rng.normal(size=3); dc /= np.linalg.norm(dc)
    width = 10.0 ** rng.uniform(-4, -0.5)
    d = dc[None] + width * rng.uniform(-1, 1, size=(64, 3))
    if kind != "raw":
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(f32)
    if kind == "camera":
        o = np.broadcast_to(C, (64, 3)).astype(f32).copy()
    else:                                                   # rays that end in C (shadow rays towards a light), plus some noise
        s = rng.uniform(0.5, 50.0, size=(64, 1))
        noise = rng.choice([0.0, 1e-6, 1e-3]) * rng.normal(size=(64, 3))
        o = (C[None].astype(np.float64) - s * d.astype(np.float64) + noise).astype(f32)
    return C, o, d


def _triangles(rng, o, d, n):
    """Triangles placed around points ON the bundle's rays, so that rays pass near (or exactly through) edges and vertices."""
    r = rng.integers(0, 64, size=n)
    t = rng.uniform(0.05, 40.0, size=(n, 1)) * rng.choice([1.0, -0.2], size=(n, 1), p=[0.9, 0.1])
    X = o[r].astype(np.float64) + t * d[r].astype(np.float64)
    size = 10.0 ** rng.uniform(-3, 1, size=(n, 1))
    a = rng.normal(size=(n, 3)) * size; b = rng.normal(size=(n, 3)) * size
    mode = rng.integers(0, 5, size=n)
    bu = rng.uniform(-1.5, 2.5, size=(n, 1)); bv = rng.uniform(-1.5, 2.5, size=(n, 1))
    bu[mode == 1] = 0.0; bv[mode == 2] = 0.0                 # X on an edge
    bu[mode == 3] = 0.0; bv[mode == 3] = 0.0                 # X on the vertex v0
    v0 = X - bu * a - bv * b
    deg = rng.random(n) < 0.03
    b[deg] = a[deg] * rng.uniform(-2, 2, size=(deg.sum(), 1))   # degenerate (zero area)
    return v0.astype(f32), a.astype(f32), b.astype(f32)


@pytest.mark.parametrize("kind", ["camera", "shadow", "raw"])
def test_culled_triangles_are_never_hit(kind):
    rng = np.random.default_rng({"camera": 1, "shadow": 2, "raw": 3}[kind])
    culled_p = culled_i = not_hit = total = 0
    for it in range(300):
        C, o, d = _bundle(rng, kind)
        v0, e1, e2 = _triangles(rng, o, d, 512)
        ol, oh, dl, dh = cm.bundle_boxes(o, d)
        delta, ok = cm.pencil_delta(o, d, C)
        for cull in (True, False):
            hit = cm.tri_accepts(o, d, v0, e1, e2, cull).any(0)
            mi = cm.interval_misses(ol, oh, dl, dh, cull, v0, e1, e2)
            assert not (mi & hit).any(), (kind, it, cull, "interval")
            if ok:
                mp = cm.pencil_misses(C, delta, ol, oh, dl, dh, cull, v0, e1, e2)
                assert not (mp & hit).any(), (kind, it, cull, "pencil", np.nonzero(mp & hit)[0][:5])
                culled_p += int((mp & ~hit).sum())
            culled_i += int((mi & ~hit).sum()); not_hit += int((~hit).sum()); total += hit.size
    # the culling must also be worth something on these near-miss-heavy sets (a sanity floor, not a performance claim)
    assert culled_p > 0.3 * not_hit or kind == "raw"
    assert culled_i > 0


def test_wrong_apex_only_loosens_the_culling():
    """The apex is a hint: rays that do not pass near it give a large delta, never a wrong answer."""
    rng = np.random.default_rng(7)
    for it in range(100):
        C, o, d = _bundle(rng, "shadow")
        wrong = (C + rng.normal(size=3).astype(f32) * f32(5.0)).astype(f32)
        v0, e1, e2 = _triangles(rng, o, d, 512)
        ol, oh, dl, dh = cm.bundle_boxes(o, d)
        delta, ok = cm.pencil_delta(o, d, wrong)
        if not ok:
            continue
        for cull in (True, False):
            hit = cm.tri_accepts(o, d, v0, e1, e2, cull).any(0)
            mp = cm.pencil_misses(wrong, delta, ol, oh, dl, dh, cull, v0, e1, e2)
            assert not (mp & hit).any(), (it, cull)
