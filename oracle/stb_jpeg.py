"""Baseline-JPEG decoder restating stb_image's algorithm (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

The reference loads bitmap textures with `stbi_load(path, &w, &h, &channels, 0)` (scene/texture/bitmap.hpp:11-37).
stb_image is a third-party dependency that is ABSENT from /root/reference: CMakeLists.txt:17-21 fetches
https://github.com/nothings/stb at `master` (unpinned) at configure time.  This file restates the published algorithm of
stb_image.h v2.2x-2.30 (`stbi__jpeg_*`, public domain / MIT) for the one kind of file the reference's scenes use —
baseline sequential JPEG (SOF0), 8-bit, Huffman — in plain Python integers:

  * `stbi__jpeg_decode_block`: DC prediction, run/size AC decoding, dequantisation `(short)(coef * dequant[zig])`
  * `stbi__idct_block`: the 12-bit fixed-point integer IDCT (`stbi__f2f(x) = (int)(x * 4096 + 0.5)`), column pass keeping two
    extra bits (`>> 10` after `+ 512`), row pass `>> 17` after `+ 65536 + (128 << 17)`, clamp to 0..255; the column shortcut
    for an all-zero AC column (`dcterm = d[0] * 4`).  stb_image's SSE2/NEON IDCT is documented there as bit-identical.
  * `stbi__resample_row_generic` / `_v_2` / `_h_2` / `_hv_2` for subsampled chroma
  * `stbi__YCbCr_to_RGB_row`: 20-bit fixed point with `stbi__float2fixed(x) = ((int)(x * 4096.0f + 0.5f)) << 8` and the
    `& 0xffff0000` on the Cb term of green (the "reduced precision" form its SIMD twin reproduces).

Parity pin: the reference's own render `outputs/textures.png` of scenes/hw12/scene4 (tests/golden/ref_outputs/textures.npz)
— every pixel of the bitmap quad depends on the decoded bytes (tests/test_reference_outputs.py).
"""
from __future__ import annotations

import struct

import numpy as np

DEZIGZAG = [
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
]


def _f2f(x: float) -> int:
    # stbi__f2f: ((int) (((x) * 4096 + 0.5))) with x a float literal: float * int -> float (exact), + 0.5 in double, truncation
    return int(float(np.float32(x) * np.float32(4096.0)) + 0.5)


def _float2fixed(x: float) -> int:
    # stbi__float2fixed: (((int) ((x) * 4096.0f + 0.5f)) << 8), all float
    return int(np.float32(np.float32(x) * np.float32(4096.0)) + np.float32(0.5)) << 8


_C = {k: _f2f(v) for k, v in dict(
    a=0.5411961, b=-1.847759065, c=0.765366865, d=1.175875602, e=0.298631336, f=2.053119869, g=3.072711026,
    h=1.501321110, i=-0.899976223, j=-2.562915447, k=-1.961570560, l=-0.390180644).items()}


def _idct_1d(s0, s1, s2, s3, s4, s5, s6, s7):
    c = _C
    p2, p3 = s2, s6
    p1 = (p2 + p3) * c["a"]
    t2 = p1 + p3 * c["b"]
    t3 = p1 + p2 * c["c"]
    p2, p3 = s0, s4
    t0 = (p2 + p3) * 4096
    t1 = (p2 - p3) * 4096
    x0, x3, x1, x2 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = s7, s5, s3, s1
    p3, p4, p1, p2 = t0 + t2, t1 + t3, t0 + t3, t1 + t2
    p5 = (p3 + p4) * c["d"]
    t0, t1, t2, t3 = t0 * c["e"], t1 * c["f"], t2 * c["g"], t3 * c["h"]
    p1 = p5 + p1 * c["i"]
    p2 = p5 + p2 * c["j"]
    p3 = p3 * c["k"]
    p4 = p4 * c["l"]
    t3 += p1 + p4
    t2 += p2 + p3
    t1 += p2 + p4
    t0 += p1 + p3
    return x0, x1, x2, x3, t0, t1, t2, t3


def _clamp(x: int) -> int:
    return 0 if x < 0 else 255 if x > 255 else x


def idct_block(d):
    """stbi__idct_block: 64 dequantised coefficients (natural order) -> 64 bytes, row-major."""
    val = [0] * 64
    for i in range(8):
        if not (d[i + 8] or d[i + 16] or d[i + 24] or d[i + 32] or d[i + 40] or d[i + 48] or d[i + 56]):
            dc = d[i] * 4
            for r in range(8):
                val[i + 8 * r] = dc
        else:
            x0, x1, x2, x3, t0, t1, t2, t3 = _idct_1d(*(d[i + 8 * r] for r in range(8)))
            x0 += 512; x1 += 512; x2 += 512; x3 += 512
            val[i] = (x0 + t3) >> 10
            val[i + 56] = (x0 - t3) >> 10
            val[i + 8] = (x1 + t2) >> 10
            val[i + 48] = (x1 - t2) >> 10
            val[i + 16] = (x2 + t1) >> 10
            val[i + 40] = (x2 - t1) >> 10
            val[i + 24] = (x3 + t0) >> 10
            val[i + 32] = (x3 - t0) >> 10
    out = [0] * 64
    bias = 65536 + (128 << 17)
    for r in range(8):
        v = val[8 * r: 8 * r + 8]
        x0, x1, x2, x3, t0, t1, t2, t3 = _idct_1d(*v)
        x0 += bias; x1 += bias; x2 += bias; x3 += bias
        o = 8 * r
        out[o + 0] = _clamp((x0 + t3) >> 17)
        out[o + 7] = _clamp((x0 - t3) >> 17)
        out[o + 1] = _clamp((x1 + t2) >> 17)
        out[o + 6] = _clamp((x1 - t2) >> 17)
        out[o + 2] = _clamp((x2 + t1) >> 17)
        out[o + 5] = _clamp((x2 - t1) >> 17)
        out[o + 3] = _clamp((x3 + t0) >> 17)
        out[o + 4] = _clamp((x3 - t0) >> 17)
    return out


class _Huff:
    def __init__(self, counts, symbols):
        self.lookup = {}
        code = 0
        k = 0
        for length in range(1, 17):
            for _ in range(counts[length - 1]):
                self.lookup[(length, code)] = symbols[k]
                code += 1
                k += 1
            code <<= 1


class _Bits:
    """Entropy-coded segment reader: 0xFF00 -> 0xFF, stops feeding at a marker (zeros after it, as stb does)."""

    def __init__(self, data: bytes, pos: int):
        self.d, self.p, self.acc, self.n, self.marker = data, pos, 0, 0, None

    def _fill(self):
        if self.marker is not None:
            b = 0
        else:
            b = self.d[self.p] if self.p < len(self.d) else 0
            self.p += 1
            if b == 0xFF:
                c = self.d[self.p] if self.p < len(self.d) else 0
                while c == 0xFF:
                    self.p += 1
                    c = self.d[self.p] if self.p < len(self.d) else 0
                self.p += 1
                if c != 0:
                    self.marker = c
                    b = 0
        self.acc = (self.acc << 8) | b
        self.n += 8

    def bit(self) -> int:
        if self.n == 0:
            self._fill()
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k: int) -> int:
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def decode(self, h: _Huff) -> int:
        code = 0
        for length in range(1, 17):
            code = (code << 1) | self.bit()
            s = h.lookup.get((length, code))
            if s is not None:
                return s
        raise ValueError("bad huffman code")

    def extend_receive(self, n: int) -> int:
        # stbi__extend_receive: n bits, values with a leading 0 bit are negative: v - (2^n - 1)
        if n == 0:
            return 0
        v = self.bits(n)
        return v if v >= (1 << (n - 1)) else v - (1 << n) + 1

    def restart(self):
        """End of a restart interval: drop the remaining bits and step over the RSTn marker (stbi__jpeg_reset)."""
        if self.marker is None and self.d[self.p: self.p + 1] == b"\xff" and 0xD0 <= self.d[self.p + 1] <= 0xD7:
            self.p += 2
        self.acc = self.n = 0
        self.marker = None


def _short(x: int) -> int:
    x &= 0xFFFF
    return x - 0x10000 if x & 0x8000 else x


def _ycbcr_row(y, cb, cr):
    R, G, B = _float2fixed(1.40200), _float2fixed(0.71414), _float2fixed(0.34414)
    B2 = _float2fixed(1.77200)
    out = np.empty((len(y), 3), np.uint8)
    for i in range(len(y)):
        yf = (int(y[i]) << 20) + (1 << 19)
        c_r = int(cr[i]) - 128
        c_b = int(cb[i]) - 128
        r = yf + c_r * R
        g = yf + c_r * -G + (((c_b * -B) >> 16) << 16)          # (cb * -fixed) & 0xffff0000 in two's complement
        b = yf + c_b * B2
        out[i, 0] = _clamp(r >> 20)
        out[i, 1] = _clamp(g >> 20)
        out[i, 2] = _clamp(b >> 20)
    return out


def _resample(comp_rows, hs, vs, width, j):
    """stb's per-output-row chroma resampling for expansion factors (hs, vs) in {1, 2}; generic nearest otherwise."""
    def row(k):
        return comp_rows[min(max(k, 0), len(comp_rows) - 1)]
    if hs == 1 and vs == 1:
        return row(j)[:width]
    if hs == 1 and vs == 2:                                                       # stbi__resample_row_v_2
        near, far = row(j >> 1), row((j >> 1) + (1 if j & 1 else -1))
        return [(3 * int(a) + int(b) + 2) >> 2 for a, b in zip(near, far)][:width]
    w_in = (width + hs - 1) // hs
    if hs == 2 and vs == 1:                                                       # stbi__resample_row_h_2
        inp = [int(x) for x in row(j)[:w_in]]
        if w_in == 1:
            return [inp[0], inp[0]][:width]
        out = [inp[0], (inp[0] * 3 + inp[1] + 2) >> 2]
        for i in range(1, w_in - 1):
            n = 3 * inp[i] + 2
            out += [(n + inp[i - 1]) >> 2, (n + inp[i + 1]) >> 2]
        out += [(inp[w_in - 2] * 3 + inp[w_in - 1] + 2) >> 2, inp[w_in - 1]]
        return out[:width]
    if hs == 2 and vs == 2:                                                       # stbi__resample_row_hv_2
        near = [int(x) for x in row(j >> 1)[:w_in]]
        far = [int(x) for x in row((j >> 1) + (1 if j & 1 else -1))[:w_in]]
        if w_in == 1:
            v = (3 * near[0] + far[0] + 2) >> 2
            return [v, v][:width]
        t1 = 3 * near[0] + far[0]
        out = [(t1 + 2) >> 2]
        for i in range(1, w_in):
            t0, t1 = t1, 3 * near[i] + far[i]
            out += [(3 * t0 + t1 + 8) >> 4, (3 * t1 + t0 + 8) >> 4]
        out.append((t1 + 2) >> 2)
        return out[:width]
    src = row(j // vs)                                                            # stbi__resample_row_generic
    return [src[i // hs] for i in range(width)]


def decode(data: bytes) -> np.ndarray:
    """uint8 [h][w][channels_in_file] exactly as stbi_load(..., req_comp = 0) returns it (1 or 3 channels)."""
    if data[:2] != b"\xff\xd8":
        raise ValueError("not a JPEG")
    pos = 2
    dequant = {}
    huff_dc, huff_ac = {}, {}
    restart = 0
    comps = None
    width = height = 0
    adobe_transform = -1
    jfif = False
    while True:
        while data[pos] != 0xFF:
            pos += 1
        while data[pos] == 0xFF:
            pos += 1
        m = data[pos]
        pos += 1
        if m == 0xD9:
            raise ValueError("no scan")
        if m in (0xC1, 0xC2):
            raise ValueError("only baseline (SOF0) JPEG is restated here")
        L = struct.unpack(">H", data[pos:pos + 2])[0]
        seg = data[pos + 2: pos + L]
        if m == 0xDB:
            q = 0
            while q < len(seg):
                p, t = seg[q] >> 4, seg[q] & 15
                q += 1
                tbl = [0] * 64
                for i in range(64):
                    if p:
                        tbl[DEZIGZAG[i]] = (seg[q] << 8) | seg[q + 1]
                        q += 2
                    else:
                        tbl[DEZIGZAG[i]] = seg[q]
                        q += 1
                dequant[t] = tbl
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1: q + 17])
                n = sum(counts)
                syms = list(seg[q + 17: q + 17 + n])
                (huff_ac if tc else huff_dc)[th] = _Huff(counts, syms)
                q += 17 + n
        elif m == 0xC0:
            prec, height, width, n = struct.unpack(">BHHB", seg[:6])
            if prec != 8 or n not in (1, 3):
                raise ValueError("unsupported JPEG layout")
            comps = [dict(id=seg[6 + 3 * k], h=seg[7 + 3 * k] >> 4, v=seg[7 + 3 * k] & 15, tq=seg[8 + 3 * k]) for k in range(n)]
        elif m == 0xDD:
            restart = struct.unpack(">H", seg[:2])[0]
        elif m == 0xE0 and seg[:5] == b"JFIF\0":
            jfif = True
        elif m == 0xEE and seg[:6] == b"Adobe\0":
            adobe_transform = seg[11]
        elif m == 0xDA:
            ns = seg[0]
            order = []
            for k in range(ns):
                cid, tbl = seg[1 + 2 * k], seg[2 + 2 * k]
                idx = next(i for i, c in enumerate(comps) if c["id"] == cid)
                comps[idx]["hd"], comps[idx]["ha"] = tbl >> 4, tbl & 15
                order.append(idx)
            pos += L
            break
        pos += L
    hmax = max(c["h"] for c in comps)
    vmax = max(c["v"] for c in comps)
    mcu_w, mcu_h = 8 * hmax, 8 * vmax
    mcux, mcuy = (width + mcu_w - 1) // mcu_w, (height + mcu_h - 1) // mcu_h
    for c in comps:
        c["w2"], c["h2"] = mcux * c["h"] * 8, mcuy * c["v"] * 8
        c["plane"] = np.zeros((c["h2"], c["w2"]), np.uint8)
        c["pred"] = 0
    br = _Bits(data, pos)

    def block(c, bx, by):
        coef = [0] * 64
        dq = dequant[c["tq"]]
        t = br.decode(huff_dc[c["hd"]])
        diff = br.extend_receive(t) if t else 0
        c["pred"] += diff
        coef[0] = _short(c["pred"] * dq[0])
        k = 1
        while k < 64:
            rs = br.decode(huff_ac[c["ha"]])
            s, r = rs & 15, rs >> 4
            if s == 0:
                if rs != 0xF0:
                    break
                k += 16
            else:
                k += r
                zig = DEZIGZAG[k]
                k += 1
                coef[zig] = _short(br.extend_receive(s) * dq[zig])
        c["plane"][by: by + 8, bx: bx + 8] = np.asarray(idct_block(coef), np.uint8).reshape(8, 8)

    todo = restart if restart else 0x7FFFFFFF
    if len(order) == 1:                                          # non-interleaved single-component scan
        c = comps[order[0]]
        w = (((width * c["h"] + hmax - 1) // hmax) + 7) >> 3
        h = (((height * c["v"] + vmax - 1) // vmax) + 7) >> 3
        for j in range(h):
            for i in range(w):
                block(c, i * 8, j * 8)
                todo -= 1
                if todo <= 0:
                    br.restart()
                    for cc in comps:
                        cc["pred"] = 0
                    todo = restart
    else:
        for j in range(mcuy):
            for i in range(mcux):
                for idx in order:
                    c = comps[idx]
                    for y in range(c["v"]):
                        for x in range(c["h"]):
                            block(c, (i * c["h"] + x) * 8, (j * c["v"] + y) * 8)
                todo -= 1
                if todo <= 0:
                    br.restart()
                    for cc in comps:
                        cc["pred"] = 0
                    todo = restart
    if len(comps) == 1:
        return np.ascontiguousarray(comps[0]["plane"][:height, :width, None])
    is_rgb = adobe_transform == 0 and not jfif or [c["id"] for c in comps] == [ord("R"), ord("G"), ord("B")]
    out = np.zeros((height, width, 3), np.uint8)
    for j in range(height):
        rows = []
        for c in comps:
            hs, vs = hmax // c["h"], vmax // c["v"]
            real_h = (height * c["v"] + vmax - 1) // vmax            # stb clamps the far row at the component's own height
            rows.append(_resample(c["plane"][:real_h], hs, vs, width, j))
        out[j] = np.stack(rows, axis=1) if is_rgb else _ycbcr_row(rows[0], rows[1], rows[2])
    return out


_cache: dict = {}


def load(path: str) -> np.ndarray:
    """decode() of a file; one decode per distinct file content and process (pure Python: seconds for a 540x360 picture)."""
    import hashlib

    with open(path, "rb") as f:
        data = f.read()
    key = hashlib.sha256(data).digest()
    if key not in _cache:
        _cache[key] = decode(data)
    return _cache[key].copy()
