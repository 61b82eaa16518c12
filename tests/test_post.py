"""CPU tests of the post-processing restatements in the oracle (SVGF_Merge.glsl:24-52,
StaticBlender.glsl:26-61, ToneMappingGamma.glsl:17-54, SSBOBuffer::save_as_image
src/SSBOBuffer.h:41-62) against float64 transcriptions of the shader text, and of the
library's PNG writer against a decoder written here."""
import ctypes as C
import struct
import zlib

import numpy as np


def test_merge_matches_transcription(oracle_mod):
    rng = np.random.default_rng(3)
    n = 4096
    di, idi = rng.random((n, 3), np.float32) * 4, rng.random((n, 3), np.float32) * 4
    al = rng.random((n, 3), np.float32)
    di[5, 1] = np.nan
    idi[9, 2] = np.inf
    al[11, 0] = -np.inf
    di[12] = 0.0; idi[12] = 0.0; al[12, 0] = np.inf  # 0 * inf = NaN
    got = oracle_mod.post_merge(di, idi, al).reshape(n, 3)
    with np.errstate(invalid="ignore"):
        want = ((di + idi) * al).astype(np.float32)
    bad = ~np.isfinite(want).all(axis=1)
    want[bad] = 10000.0
    assert np.array_equal(got, want)
    assert sorted(np.nonzero(bad)[0].tolist()) == [5, 9, 11, 12]


def test_tonemap_matches_float64(oracle_mod):
    x = np.concatenate([np.linspace(0, 4, 4001), [10000.0, 1e-6, 0.18]]).astype(np.float32)
    got = oracle_mod.post_tonemap(x).astype(np.float64)
    c = x.astype(np.float64)
    aces = (c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14)
    want = np.where(aces > 0, aces ** 0.45, 0.0)
    assert np.max(np.abs(got - want)) < 2e-6
    assert got[0] == 0.0


def test_static_blend_semantics(oracle_mod):
    rng = np.random.default_rng(5)
    n = 1000
    hist_c = np.zeros((n, 3), np.float32)
    hist_m = np.zeros((n, 2), np.float32)
    frames = [rng.random((n, 3), np.float32) for _ in range(4)]
    frames[2][7] = 500.0  # a firefly at frame 3: rejected (3 sigma, sigma >= 1)
    means = np.zeros((n, 3), np.float64)
    for f, col in enumerate(frames, start=1):
        hc, hm = oracle_mod.post_static_blend(f, col, hist_c, hist_m)
        hist_c, hist_m = hc.reshape(n, 3), hm.reshape(n, 2)
        if f == 2:
            before_firefly = hist_c[7].copy()
        if f == 3:
            assert np.array_equal(hist_c[7], before_firefly)
    # pixels without rejection hold the running mean
    want = np.mean(np.stack(frames), axis=0)
    ok = np.ones(n, bool); ok[7] = False
    assert np.max(np.abs(hist_c[ok] - want[ok])) < 1e-6
    # frame 1 replaces the history outright even if it is garbage
    hc, hm = oracle_mod.post_static_blend(1, frames[0], np.full((n, 3), 1e9, np.float32), np.full((n, 2), 1e9, np.float32))
    assert np.allclose(hc.reshape(n, 3), frames[0] , atol=0, rtol=0) or np.max(np.abs(hc.reshape(n, 3) - frames[0])) == 0


def test_quantize_image(oracle_mod):
    from raytracing_amd import _capi
    w, h, c = 5, 3, 3
    data = np.linspace(-0.2, 1.3, w * h * c).astype(np.float32)
    data[4] = np.nan
    want = oracle_mod.quantize_image(data, w, h, c)
    # transcription: min(255, int(x*255)) as a byte, rows flipped
    v = np.where(np.isnan(data), 0, np.minimum(255, np.trunc(np.nan_to_num(data) * np.float32(255)).astype(np.int64)))
    ref = (v & 0xFF).astype(np.uint8).reshape(h, w, c)[::-1]
    assert np.array_equal(want, ref)
    got = np.zeros(w * h * c, np.uint8)
    _capi.check(_capi.lib.pt_host_quantize_image(data.ctypes.data, w, h, c, got.ctypes.data))
    assert np.array_equal(got.reshape(h, w, c), want)


def decode_png(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) == crc
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype = hdr[:4]
    c = {0: 1, 2: 3, 6: 4}[ctype]
    assert depth == 8
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * c)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, c)


def test_png_writer_roundtrip(tmp_path):
    from raytracing_amd import _capi
    rng = np.random.default_rng(1)
    for (w, h, c) in [(1, 1, 1), (7, 5, 3), (300, 200, 4), (640, 480, 3)]:  # the last needs several stored blocks
        px = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        path = str(tmp_path / ("t_%d_%d_%d.png" % (w, h, c)))
        _capi.check(_capi.lib.pt_host_write_png(path.encode(), w, h, c, px.ctypes.data))
        assert np.array_equal(decode_png(path), px)
    assert _capi.lib.pt_host_write_png(b"/nonexistent_dir/x.png", 2, 2, 3, px.ctypes.data) != 0
    assert _capi.lib.pt_host_write_png(str(tmp_path / "bad.png").encode(), 2, 2, 2, px.ctypes.data) != 0
