"""numpy-float32 model of the two pieces of trace.hip.hpp whose soundness rests on floating-point arguments:

* `tri_accepts`      the per-ray test of tri_step / test_triangle (= triangle_packet::intersect, kd_tree_simd.hpp:25-60),
                     same operations in the same order, float32, no FMA;
* `pencil_misses`    the apex ("pencil") bundle culling: a triangle is dropped for a whole bundle of rays whose lines pass
                     within `delta` of a common point C (camera rays: the camera; shadow rays: the light), when linear
                     bounds over the bundle's direction box, widened by rounding-error margins, prove that the per-ray test
                     fails for every ray;
* `interval_misses`  the interval-arithmetic culling of generic bundles (sound by monotone rounding alone).

The model exists so that the soundness property "culled => no ray of the bundle is accepted" can be hammered on the CPU with
millions of random and adversarial cases (tests/test_bundle_cull_model.py).  It is test infrastructure: the product's
culling lives in simd-raytracer_amd/csrc/trace.hip.hpp and is checked end to end by the GPU parity tests.
"""
import numpy as np

f32 = np.float32
EPS24 = f32(2.0 ** -24)
K_T = f32(2.0 ** -20)          # rounding-error margin per unit of |operand products| (derived bound: 6.1 * 2^-24 twice)
K_DET = f32(2.0 ** -21)        # relative slack on the determinant in the u <= 1 and u + v <= 1 tests (derived: 4 * 2^-24)
ABS_SLACK = f32(1e-30)


def _f(x):
    return np.asarray(x, dtype=f32)


def tri_accepts(o, d, v0, e1, e2, cull, eps=f32(1e-6)):
    """[rays, tris] bool: the exact per-ray acceptance (without the running `t < best.t`).  o, d: [R, 3]; v0, e1, e2: [K, 3]."""
    o = _f(o)[:, None, :]; d = _f(d)[:, None, :]
    v0 = _f(v0)[None]; e1 = _f(e1)[None]; e2 = _f(e2)[None]
    with np.errstate(all="ignore"):
        pvx = d[..., 1] * e2[..., 2] - d[..., 2] * e2[..., 1]
        pvy = d[..., 2] * e2[..., 0] - d[..., 0] * e2[..., 2]
        pvz = d[..., 0] * e2[..., 1] - d[..., 1] * e2[..., 0]
        det = (e1[..., 0] * pvx + e1[..., 1] * pvy) + e1[..., 2] * pvz
        m = (eps <= det) if cull else (eps <= np.abs(det))
        inv = f32(1.0) / det
        tvx = o[..., 0] - v0[..., 0]; tvy = o[..., 1] - v0[..., 1]; tvz = o[..., 2] - v0[..., 2]
        u = ((tvx * pvx + tvy * pvy) + tvz * pvz) * inv
        m &= (f32(0) <= u) & (u <= f32(1))
        qx = tvy * e1[..., 2] - tvz * e1[..., 1]
        qy = tvz * e1[..., 0] - tvx * e1[..., 2]
        qz = tvx * e1[..., 1] - tvy * e1[..., 0]
        v = ((d[..., 0] * qx + d[..., 1] * qy) + d[..., 2] * qz) * inv
        m &= (f32(0) <= v) & (u + v <= f32(1))
        t = ((e2[..., 0] * qx + e2[..., 1] * qy) + e2[..., 2] * qz) * inv
        m &= eps < t
    return m


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _adot(a, b):          # |a| . b  (b non-negative)
    return (np.abs(a[0]) * b[0] + np.abs(a[1]) * b[1]) + np.abs(a[2]) * b[2]


def _across(a, b):        # "absolute cross product": upper bound of |x x y| for |x| <= a, |y| <= b, componentwise
    return [a[1] * b[2] + a[2] * b[1], a[2] * b[0] + a[0] * b[2], a[0] * b[1] + a[1] * b[0]]


def pencil_delta(o, d, C):
    """Upper bound of the distance between the apex C and each ray's line (float32, as the device computes it)."""
    o = _f(o); d = _f(d); C = _f(C)
    with np.errstate(all="ignore"):
        c = [C[a] - o[:, a] for a in range(3)]
        dd = _dot([d[:, 0], d[:, 1], d[:, 2]], [d[:, 0], d[:, 1], d[:, 2]])
        k = _dot(c, [d[:, 0], d[:, 1], d[:, 2]]) / dd
        w = [c[a] - k * d[:, a] for a in range(3)]
        w1 = (np.abs(w[0]) + np.abs(w[1])) + np.abs(w[2])
        c1 = (np.abs(c[0]) + np.abs(c[1])) + np.abs(c[2])
        dl = w1 + f32(2.0 ** -19) * c1
    ok = bool(np.all(dd >= f32(1e-30)) and np.all(np.isfinite(dl)))
    return (f32(np.max(dl)) * f32(1.00001) if ok else f32(np.inf)), ok


def bundle_boxes(o, d):
    o = _f(o); d = _f(d)
    return o.min(0), o.max(0), d.min(0), d.max(0)


def pencil_misses(C, delta, ol, oh, dl, dh, all_cull, v0, e1, e2, eps=f32(1e-6)):
    """[K] bool: True = no ray of the pencil bundle can be accepted for this triangle.  Mirrors pencil_misses() of trace.hip.hpp."""
    C = _f(C); ol = _f(ol); oh = _f(oh); dl = _f(dl); dh = _f(dh); delta = f32(delta)
    v0 = _f(v0); e1 = _f(e1); e2 = _f(e2)
    V0 = [v0[:, a] for a in range(3)]; E1 = [e1[:, a] for a in range(3)]; E2 = [e2[:, a] for a in range(3)]
    with np.errstate(all="ignore"):
        # centre / radius of the direction and origin boxes, radii inflated so that centre +- radius covers the box in exact arithmetic
        dc = [(dl[a] + dh[a]) * f32(0.5) for a in range(3)]
        rd = [(dh[a] - dl[a]) * f32(0.5) * f32(1.000001) + f32(2.0 ** -22) * (np.abs(dl[a]) + np.abs(dh[a])) for a in range(3)]
        oc = [(ol[a] + oh[a]) * f32(0.5) for a in range(3)]
        ro = [(oh[a] - ol[a]) * f32(0.5) * f32(1.000001) + f32(2.0 ** -22) * (np.abs(ol[a]) + np.abs(oh[a])) for a in range(3)]
        Dm = [np.abs(dc[a]) + rd[a] for a in range(3)]
        D1 = (Dm[0] + Dm[1]) + Dm[2]
        cv = [C[a] - V0[a] for a in range(3)]
        A = _cross(E2, cv)              # un = d . A     (+ a term bounded by delta |d| |e2|)
        Bv = _cross(cv, E1)             # vn = d . Bv    (+ a term bounded by delta |d| |e1|)
        Dv = _cross(E2, E1)             # det = d . Dv
        tvc = [oc[a] - V0[a] for a in range(3)]
        TVm = [np.abs(tvc[a]) + ro[a] for a in range(3)]
        TC = [np.abs(cv[a]) for a in range(3)]
        aE1 = [np.abs(E1[a]) for a in range(3)]; aE2 = [np.abs(E2[a]) for a in range(3)]
        E21 = (aE2[0] + aE2[1]) + aE2[2]
        E11 = (aE1[0] + aE1[1]) + aE1[2]
        E1m = np.maximum(np.maximum(aE1[0], aE1[1]), aE1[2]); E2m = np.maximum(np.maximum(aE2[0], aE2[1]), aE2[2])
        # the T's of the error analysis, bounded through norms: sum_a x_a (y_b z_c + y_c z_b) <= |x|_1 |y|_1 max|z|
        TV1 = (TVm[0] + TVm[1]) + TVm[2]
        TS = TV1 + ((TC[0] + TC[1]) + TC[2])
        dD1 = (delta * D1) * f32(1.00001)
        M_un = K_T * ((TS * D1) * E2m) + dD1 * E21 + ABS_SLACK
        M_vn = K_T * ((TS * D1) * E1m) + dD1 * E11 + ABS_SLACK
        M_det = K_T * ((E11 * D1) * E2m) + ABS_SLACK
        M_tn = K_T * ((E21 * TV1) * E1m) + ABS_SLACK
        c_det = _dot(dc, Dv); r_det = _adot(Dv, rd)
        detH = (c_det + r_det) + M_det
        detL = (c_det - r_det) - M_det
        none = (detH < eps) & (all_cull | (-eps < detL))
        pos = ~(detH < eps)
        neg = (~np.bool_(all_cull)) & ~(-eps < detL)
        s = np.where(neg, f32(-1), f32(1))                   # normalise to positive determinants
        dH = np.where(neg, -detL, detH)
        c_un = s * _dot(dc, A); r_un = _adot(A, rd)
        c_vn = s * _dot(dc, Bv); r_vn = _adot(Bv, rd)
        X1 = [A[a] - Dv[a] for a in range(3)]
        X2 = [X1[a] + Bv[a] for a in range(3)]
        c_x1 = s * _dot(dc, X1); r_x1 = _adot(X1, rd)
        c_x2 = s * _dot(dc, X2); r_x2 = _adot(X2, rd)
        c_tn = -s * _dot(tvc, Dv); r_tn = _adot(Dv, ro)
        rcp = f32(1.0) / dH
        unH = (c_un + r_un) + M_un
        vnH = (c_vn + r_vn) + M_vn
        slack_d = K_DET * dH
        out = (unH < 0) & ((-unH) * rcp >= f32(1e-30))
        out |= ((c_x1 - r_x1) - (M_un + M_det) - slack_d) > 0
        out |= (vnH < 0) & ((-vnH) * rcp >= f32(1e-30))
        out |= ((c_x2 - r_x2) - ((M_un + M_vn) + M_det) - slack_d) > 0
        out |= ((c_tn + r_tn) + M_tn) < 0
        return none | (out & ~(pos & neg) & (dH < f32(1e30)))


class Iv:
    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi


def _iv_scale(a, s):
    p, q = a.lo * s, a.hi * s
    return Iv(np.minimum(p, q), np.maximum(p, q))


def _iv_mul(a, b):
    c = [a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi]
    return Iv(np.minimum(np.minimum(c[0], c[1]), np.minimum(c[2], c[3])), np.maximum(np.maximum(c[0], c[1]), np.maximum(c[2], c[3])))


def _iv_sub(a, b):
    return Iv(a.lo - b.hi, a.hi - b.lo)


def _iv_add(a, b):
    return Iv(a.lo + b.lo, a.hi + b.hi)


def interval_misses(ol, oh, dl, dh, all_cull, v0, e1, e2, eps=f32(1e-6)):
    """[K] bool: the interval-arithmetic culling of generic bundles (bundle_misses() of trace.hip.hpp)."""
    ol = _f(ol); oh = _f(oh); dl = _f(dl); dh = _f(dh)
    v0 = _f(v0); e1 = _f(e1); e2 = _f(e2)
    one = np.ones(v0.shape[0], f32)
    with np.errstate(all="ignore"):
        dx, dy, dz = (Iv(dl[a] * one, dh[a] * one) for a in range(3))
        pvx = _iv_sub(_iv_scale(dy, e2[:, 2]), _iv_scale(dz, e2[:, 1]))
        pvy = _iv_sub(_iv_scale(dz, e2[:, 0]), _iv_scale(dx, e2[:, 2]))
        pvz = _iv_sub(_iv_scale(dx, e2[:, 1]), _iv_scale(dy, e2[:, 0]))
        det = _iv_add(_iv_add(_iv_scale(pvx, e1[:, 0]), _iv_scale(pvy, e1[:, 1])), _iv_scale(pvz, e1[:, 2]))
        none = (det.hi < eps) & (all_cull | (-eps < det.lo))
        pos = ~(det.hi < eps)
        neg = (~np.bool_(all_cull)) & ~(-eps < det.lo)
        tvx, tvy, tvz = (Iv(ol[a] - v0[:, a], oh[a] - v0[:, a]) for a in range(3))
        un = _iv_add(_iv_add(_iv_mul(tvx, pvx), _iv_mul(tvy, pvy)), _iv_mul(tvz, pvz))
        qx = _iv_sub(_iv_scale(tvy, e1[:, 2]), _iv_scale(tvz, e1[:, 1]))
        qy = _iv_sub(_iv_scale(tvz, e1[:, 0]), _iv_scale(tvx, e1[:, 2]))
        qz = _iv_sub(_iv_scale(tvx, e1[:, 1]), _iv_scale(tvy, e1[:, 0]))
        vn = _iv_add(_iv_add(_iv_mul(dx, qx), _iv_mul(dy, qy)), _iv_mul(dz, qz))
        tn = _iv_add(_iv_add(_iv_scale(qx, e2[:, 0]), _iv_scale(qy, e2[:, 1])), _iv_scale(qz, e2[:, 2]))
        d_l = np.where(neg, -det.hi, det.lo); d_h = np.where(neg, -det.lo, det.hi)
        un = Iv(np.where(neg, -un.hi, un.lo), np.where(neg, -un.lo, un.hi))
        vn = Iv(np.where(neg, -vn.hi, vn.lo), np.where(neg, -vn.lo, vn.hi))
        tn_hi = np.where(neg, -tn.lo, tn.hi)
        d_l = np.maximum(d_l, eps)
        il = f32(1.0) / d_h * f32(0.999999); ih = f32(1.0) / d_l * f32(1.000001)
        u_hi = np.maximum(un.hi * il, un.hi * ih); u_lo = np.minimum(un.lo * il, un.lo * ih)
        v_hi = np.maximum(vn.hi * il, vn.hi * ih); v_lo = np.minimum(vn.lo * il, vn.lo * ih)
        t_hi = np.maximum(tn_hi * il, tn_hi * ih)
        out = (u_hi < 0) | (f32(1) < u_lo) | (v_hi < 0) | (f32(1) < u_lo + v_lo) | (t_hi <= eps)
        return none | (out & ~(pos & neg) & (d_h < f32(1e30)))
