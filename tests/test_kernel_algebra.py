"""The integer identities round 4's kernel rewrites rest on, checked exhaustively (colour terms) or on wide random and extreme inputs
(IDCT) in numpy -- no device. The GPU parity tests compare whole images with the oracle; these pin the algebra itself, for EVERY
input the kernels can meet:

* k_resample_420 / k_area_420 (lp_kernels_pixel.hip, lp_area_core.h): jdcolor.c's  y + ((FIX * (c - 128) + ONE_HALF) >> 16)  as
  "luma + upper half of a 16-bit dot product", with FIX(1.772) = 2 * 58065, FIX(1.402) = 3 * 30627 and green as
  -floor((x + 65535 - KG) / 65536); the 16-bit wrap-around add and v_sat_pk_u8_i16's clamp in place of the 32-bit add and clamp.
* k_idct's fast tile path (lp_kernels_decode.hip): the pass's rounding constant inside tmp0 / tmp1 instead of added to every output,
  d0 / d4 multiplied by q << 13, and "(o + C) >> 18, clamped" as "upper half of o + C, shifted right by two as a 16-bit value, saturated".
"""
import numpy as np


def FIX(x):
    return int(x * 65536.0 + 0.5)


def _i16(x):  # the low 16 bits of x as a signed 16-bit value (what a packed 16-bit add leaves in a half register)
    return ((np.asarray(x, np.int64) + 32768) % 65536 - 32768).astype(np.int64)


def _upper_half(x):  # bits 16..31 of a 32-bit register, read as signed 16-bit
    u = np.asarray(x, np.int64) % (1 << 32)
    return _i16(u >> 16)


def _sat_u8(x):  # v_sat_pk_u8_i16 on one half
    return np.clip(x, 0, 255)


def test_colour_terms_as_dot_products_for_every_y_cb_cr():
    cb, cr = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    KR, KB = 32768 - 128 * FIX(1.40200), 32768 - 128 * FIX(1.77200)
    KG = 32768 + 128 * FIX(0.34414) + 128 * FIX(0.71414)
    assert FIX(1.77200) == 2 * 58065 and FIX(1.40200) == 3 * 30627 and max(58065, 30627, FIX(0.34414), FIX(0.71414)) < 65536
    # jdcolor.c build_ycc_rgb_table / ycc_rgb_convert (RIGHT_SHIFT is arithmetic: floor)
    tr_ref = (FIX(1.40200) * (cr - 128) + 32768) >> 16
    tb_ref = (FIX(1.77200) * (cb - 128) + 32768) >> 16
    tg_ref = (-FIX(0.34414) * (cb - 128) - FIX(0.71414) * (cr - 128) + 32768) >> 16
    # the kernels: v_pk_mul_lo_u16 {2, 3}, v_dot2_u32_u16 with the constant as accumulator, the term = the upper half
    hs_b, hs_r = 2 * cb, 3 * cr
    assert hs_b.max() < 65536 and hs_r.max() < 65536
    tr = _upper_half(hs_r * 30627 + KR)
    tb = _upper_half(hs_b * 58065 + KB)
    zg = _upper_half(cb * FIX(0.34414) + cr * FIX(0.71414) + 65535 - KG)
    assert np.array_equal(tr, tr_ref) and np.array_equal(tb, tb_ref) and np.array_equal(-zg, tg_ref)
    for y in range(256):  # the 16-bit add / subtract of the luma byte, then the saturating pack
        assert np.array_equal(_sat_u8(_i16(y + tr)), np.clip(y + tr_ref, 0, 255))
        assert np.array_equal(_sat_u8(_i16(y + tb)), np.clip(y + tb_ref, 0, 255))
        assert np.array_equal(_sat_u8(_i16(y - zg)), np.clip(y + tg_ref, 0, 255))


def _w32(x):  # 32-bit two's-complement wrap-around
    return ((np.asarray(x, np.int64) + (1 << 31)) % (1 << 32)) - (1 << 31)


def _idct_1d(d):  # jidctint.c, one pass, 32-bit arithmetic as the kernels (and the oracle) do it; d: [..., 8]
    d0, d1, d2, d3, d4, d5, d6, d7 = [d[..., i] for i in range(8)]
    z1 = _w32((d2 + d6) * 4433)
    tmp2, tmp3 = _w32(z1 - d6 * 15137), _w32(z1 + d2 * 6270)
    tmp0, tmp1 = _w32((d0 + d4) << 13), _w32((d0 - d4) << 13)
    return _idct_1d_core(d1, d3, d5, d7, tmp0, tmp1, tmp2, tmp3)


def _idct_1d_core(d1, d3, d5, d7, tmp0, tmp1, tmp2, tmp3):
    t10, t13, t11, t12 = _w32(tmp0 + tmp3), _w32(tmp0 - tmp3), _w32(tmp1 + tmp2), _w32(tmp1 - tmp2)
    z1, z2, z3, z4 = d7 + d1, d5 + d3, d7 + d3, d5 + d1
    z5 = _w32((z3 + z4) * 9633)
    a0, a1, a2, a3 = _w32(d7 * 2446), _w32(d5 * 16819), _w32(d3 * 25172), _w32(d1 * 12299)
    z1, z2 = _w32(z1 * -7373), _w32(z2 * -20995)
    z3, z4 = _w32(z3 * -16069 + z5), _w32(z4 * -3196 + z5)
    a0, a1, a2, a3 = _w32(a0 + z1 + z3), _w32(a1 + z2 + z4), _w32(a2 + z2 + z3), _w32(a3 + z1 + z4)
    return np.stack([_w32(t10 + a3), _w32(t11 + a2), _w32(t12 + a1), _w32(t13 + a0), _w32(t13 - a0), _w32(t12 - a1), _w32(t11 - a2), _w32(t10 - a3)], axis=-1)


def _idct_1d_pre(d, tmp0, tmp1):  # idct_1d_pre: the even part's tmp0 / tmp1 arrive ready-made
    d1, d2, d3, d5, d6, d7 = d[..., 1], d[..., 2], d[..., 3], d[..., 5], d[..., 6], d[..., 7]
    z1 = _w32((d2 + d6) * 4433)
    return _idct_1d_core(d1, d3, d5, d7, tmp0, tmp1, _w32(z1 - d6 * 15137), _w32(z1 + d2 * 6270))


def _blocks(rng, n):
    """Coefficient blocks [n, 8 rows, 8 columns] the fast path can meet: AC within the int8 range (no escape), a 16-bit DC, 8-bit
    quantisers; half of them photographic (decaying), half adversarial (every value at an extreme)."""
    c = np.zeros((n, 8, 8), np.int64)
    decay = 1.0 / (1.0 + np.add.outer(np.arange(8), np.arange(8)))
    c[: n // 2] = np.rint(rng.normal(0, 60, (n // 2, 8, 8)) * decay).clip(-127, 127)
    sparse = rng.random((n - n // 2, 8, 8)) < 0.12                      # a handful of extreme coefficients per block
    c[n // 2 :] = rng.choice(np.array([-127, -1, 1, 127]), (n - n // 2, 8, 8)) * sparse
    c[:, 0, 0] = rng.integers(-2047, 2048, n)
    q = rng.integers(1, 48, (n, 8, 8))
    q[::5] = rng.integers(1, 256, (len(q[::5]), 8, 8))                  # coarse tables too (most of those blocks fall to the bound below)
    # keep the blocks a decoder can meet without 32-bit overflow in jidctint.c's own arithmetic (a reconstructed sample of a few
    # thousand at most: |sum of dequantised coefficients| / 8); beyond that the C code (64-bit JLONG), its SIMD twins and any 32-bit
    # restatement part ways anyway, whatever the formulation
    ok = np.abs(c * q).sum(axis=(1, 2)) < 30000
    return c[ok], q[ok]


def test_idct_fast_tile_path_equals_the_plain_formulation():
    rng = np.random.default_rng(4)
    c, q = _blocks(rng, 20000)
    assert len(c) > 6000
    # plain: dequantise, column pass with DESCALE(x, 11), row pass with DESCALE(x, 18) + 128, clamp (jidctint.c; oracle/jpeg_oracle.c)
    d = _w32(c * q)
    ws = np.swapaxes(_w32(_idct_1d(np.swapaxes(d, 1, 2)) + (1 << 10)) >> 11, 1, 2)  # [n, row, column]
    plain = np.clip((_w32(_idct_1d(ws) + (1 << 17)) >> 18) + 128, 0, 255)
    # fast path, column pass: d0 and d4 multiplied by q << 13 (24-bit operands, 32-bit wrap), the rounding constant inside tmp0 / tmp1
    dc = np.swapaxes(d, 1, 2)                                                      # [n, column, row index as the 1-D position]
    c0, q0, c4, q4 = c[:, 0, :], q[:, 0, :], c[:, 4, :], q[:, 4, :]                # per column: rows 0 and 4
    assert np.abs(c0).max() < (1 << 23) and (q0 << 13).max() < (1 << 23)           # v_mul_i32_i24's operand range
    t0e, d4s = _w32(c0 * (q0 << 13) + (1 << 10)), _w32(c4 * (q4 << 13))
    ws_fast = np.swapaxes(_idct_1d_pre(dc, _w32(t0e + d4s), _w32(t0e - d4s)) >> 11, 1, 2)
    assert np.array_equal(ws_fast, ws)
    # row pass: the constant inside tmp0 / tmp1; output = upper half of the sum, >> 2 as a 16-bit value, saturated to a byte
    C = (1 << 17) + (128 << 18)
    a, b = ws_fast[..., 0], ws_fast[..., 4]
    o = _idct_1d_pre(ws_fast, _w32(((a + b) << 13) + C), _w32(((a - b) << 13) + C))
    fast = np.clip(_upper_half(o) >> 2, 0, 255)
    assert np.array_equal(fast, plain)
    assert (plain == 0).any() and (plain == 255).any()  # both clamps were exercised
