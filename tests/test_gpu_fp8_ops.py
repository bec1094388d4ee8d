"""GPU: the fp8 trunk's kernels one by one through the C ABI (rsys_op_f8_quantize / rsys_op_f8_weights / rsys_op_gemm_f8) against
the numpy restatement of torchao's tensor-wise recipe (oracle/fp8.py: pinned to torch's float8 casts and torch._scaled_mm on the CPU, torchao itself being absent):
casts bit-exact, GEMMs exact on data whose products and sums are exact and to fp32 accumulation order on random data."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lib():
    from recommendersystem_amd import _lib
    return _lib.lib(), _lib


def _bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) >> 16).astype(np.uint16)


def _bf16_val(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def _dev(lib, arr):
    arr = np.ascontiguousarray(arr)
    p = C.c_void_p()
    assert lib.rsys_dev_alloc(C.byref(p), max(arr.nbytes, 64)) == 0
    assert lib.rsys_dev_h2d(p, arr.ctypes.data, arr.nbytes) == 0
    return p


def _empty(lib, nbytes, fill=0):
    p = C.c_void_p()
    assert lib.rsys_dev_alloc(C.byref(p), max(nbytes, 64)) == 0
    assert lib.rsys_dev_memset(p, fill, max(nbytes, 64)) == 0
    return p


def _get(lib, p, shape, dtype):
    out = np.empty(shape, dtype)
    assert lib.rsys_dev_d2h(out.ctypes.data, p, out.nbytes) == 0
    return out


def _segments(cols, layout, seg_cols, seg_rep=1):
    """column index arrays of the amax segments, and the destination column of every source column"""
    c = np.arange(cols)
    if layout == 1:
        u = c // seg_cols
        sg = np.where(u < seg_rep, 0, u - seg_rep + 1)
        return [c[sg == s] for s in range(sg.max() + 1)], c
    if layout == 2:
        isb = (c >> 4) & 1
        dest = np.where(isb == 1, cols // 2, 0) + (c >> 5) * 16 + (c & 15)
        return [c[isb == 0], c[isb == 1]], dest
    return [c], c


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("rows,cols,layout,seg_cols,seg_rep", [(300, 512, 0, 0, 1), (257, 1536, 1, 512, 1), (130, 2816, 2, 0, 1), (64, 192, 1, 64, 1),
                                                                (200, 1024, 1, 256, 2), (100, 640, 1, 128, 3)])
def test_quantize_is_bit_exact(fmt, rows, cols, layout, seg_cols, seg_rep):
    from oracle import fp8
    lib, L = _lib()
    rng = np.random.default_rng(rows + cols + fmt)
    x = rng.standard_normal((rows, cols)).astype(np.float32) * np.exp(rng.uniform(-6, 3, (1, cols))).astype(np.float32)
    x[rng.random((rows, cols)) < 0.05] = 0.0
    xb = _bf16_bits(x); xv = _bf16_val(xb)
    ld = cols + 16
    src = np.zeros((rows, ld), np.uint16); src[:, :cols] = xb
    d_src = _dev(lib, src); d_dst = _empty(lib, rows * cols, 0x55); d_amax = _empty(lib, 64 * 32 * 4); d_desc = _empty(lib, 128)
    wam = np.array([0.7, 1.9, 0.031, 5.0], np.float32)
    d_wam = _dev(lib, wam)
    segs, dest = _segments(cols, layout, seg_cols, seg_rep)
    mode = 2 if (fmt == 1 and layout != 0) else 1
    n_w = len(segs) if mode == 2 else 3
    w_rep = seg_rep
    rc = lib.rsys_op_f8_quantize(d_src, ld, rows, cols, fmt, layout, seg_cols, seg_rep, d_dst, cols, d_amax, d_desc, d_wam, n_w, w_rep, mode)
    assert rc == 0, L.last_error()
    amax = _get(lib, d_amax, (64, 32), np.float32).max(0)      # an amax slot is 64 shards; element [shard][segment]
    got = _get(lib, d_dst, (rows, cols), np.uint8)
    desc = _get(lib, d_desc, (32,), np.float32)
    want = np.zeros((rows, cols), np.uint8)
    scales = []
    for si, cidx in enumerate(segs):
        am = np.abs(xv[:, cidx]).max()
        assert amax[si] == am, (si, amax[si], am)
        s = fp8.scale_of(am, fmt); scales.append(s)
        want[:, dest[cidx]] = fp8.encode_fp8(fp8.round_fp8(xv[:, cidx] * s, fmt), fmt)
    # (-0.0 and +0.0 are both zero: compare decoded values where the codes differ only in the sign of zero)
    diff = got != want
    assert not (diff & ((got & 0x7F) != 0)).any() and not (diff & ((want & 0x7F) != 0)).any(), int(diff.sum())
    sw = fp8.scale_of(wam, fp8.E4M3)
    if mode == 1:
        for u in range(n_w - 1 + w_rep):
            assert desc[u] == fp8.descale(scales[0], sw[0 if u < w_rep else u - w_rep + 1]), u
    else:
        cj = [fp8.descale(scales[j], sw[j]) for j in range(len(segs))]
        assert desc[0] == cj[-1]
        for j in range(len(segs) - 1):
            assert desc[16 + j] == np.float32(cj[j]) / np.float32(cj[j + 1])
    for p in (d_src, d_dst, d_amax, d_desc, d_wam):
        lib.rsys_dev_free(p)


@pytest.mark.parametrize("rows,cols,layout,seg_rows,seg_rep", [(512, 512, 0, 0, 1), (1536, 512, 1, 512, 1), (2816, 512, 2, 0, 1), (512, 1408, 0, 0, 1), (96, 68, 2, 0, 1),
                                                                (1024, 512, 1, 256, 2)])
def test_weight_copies_are_bit_exact(rows, cols, layout, seg_rows, seg_rep):
    from oracle import fp8
    lib, L = _lib()
    rng = np.random.default_rng(rows * 3 + cols)
    w = (rng.standard_normal((rows, cols)) * np.exp(rng.uniform(-3, 1, (rows, 1)))).astype(np.float32) * 0.05
    d_w = _dev(lib, w); d_dst = _empty(lib, rows * cols, 0x33); d_t = _empty(lib, rows * cols, 0x33); d_amax = _empty(lib, 16)
    rc = lib.rsys_op_f8_weights(d_w, cols, rows, cols, layout, seg_rows, seg_rep, d_amax, d_dst, d_t, rows)
    assert rc == 0, L.last_error()
    amax = _get(lib, d_amax, (4,), np.float32)
    got = _get(lib, d_dst, (rows, cols), np.uint8); got_t = _get(lib, d_t, (cols, rows), np.uint8)
    r = np.arange(rows)
    if layout == 1:
        u = r // seg_rows
        seg = np.where(u < seg_rep, 0, u - seg_rep + 1); tcol = r
    elif layout == 2:
        seg = (r >> 4) & 1; tcol = np.where(seg == 1, rows // 2, 0) + (r >> 5) * 16 + (r & 15)
    else:
        seg = np.zeros(rows, int); tcol = r
    want = np.zeros((rows, cols), np.uint8)
    for s in range(seg.max() + 1):
        am = np.abs(w[seg == s]).max()
        assert amax[s] == am
        want[seg == s] = fp8.encode_fp8(fp8.round_fp8(w[seg == s] * fp8.scale_of(am, fp8.E4M3), fp8.E4M3), fp8.E4M3)
    same = lambda a, b: not ((a != b) & (((a & 0x7F) != 0) | ((b & 0x7F) != 0))).any()
    assert same(got, want)
    want_t = np.zeros((cols, rows), np.uint8); want_t[:, tcol] = want.T
    assert same(got_t, want_t)
    for p in (d_w, d_dst, d_t, d_amax):
        lib.rsys_dev_free(p)


def _grid_values(fmt, rng, shape, small):
    """random fp8 codes; `small`: values whose products and sums stay exact in fp32 (integers up to 4 times a power of two)"""
    from oracle import fp8
    if small:
        v = rng.integers(-4, 5, shape).astype(np.float32) * np.float32(0.25)
        return fp8.encode_fp8(v, fmt), v
    codes = rng.integers(0, 256, shape).astype(np.uint8)
    v = fp8.decode_fp8(codes, fmt)
    bad = ~np.isfinite(v) | (np.abs(v) > fp8.fmax(fmt))
    codes[bad] = 0; v[bad] = 0.0
    if fmt == 1:                                      # keep e5m2 magnitudes in a range whose products stay finite in fp32
        big = np.abs(v) > 64.0
        codes[big] = 0; v[big] = 0.0
    return codes, v


def run_gemm_f8(M, N, K, a_fmt=0, c_f32=True, desc=(1.0,), seg_cols=0, alt=0, kb=(), small=True, seed=0):
    """kb: K tiles (of 128) at which a new K segment begins; desc then holds one descale per segment"""
    lib, L = _lib()
    rng = np.random.default_rng(seed)
    Ac, Av = _grid_values(a_fmt, rng, (M, K), small)
    Bc, Bv = _grid_values(0, rng, (N, K), small)
    if small:   # asymmetric rows / columns: catches transposed or permuted fragments
        Av[0, :] = (np.arange(K) % 5 - 2) * 0.5; Bv[:, 0] = (np.arange(N) % 7 - 3) * 0.5
        from oracle import fp8
        Ac = fp8.encode_fp8(Av, a_fmt); Bc = fp8.encode_fp8(Bv, 0)
    d = np.zeros(32, np.float32); d[:len(desc)] = desc
    if kb:
        cj = list(desc[:len(kb) + 1])
        d[:] = 0; d[0] = cj[-1]
        for j in range(len(kb)):
            d[16 + j] = np.float32(cj[j]) / np.float32(cj[j + 1])
    kbs = list(kb) + [0] * (3 - len(kb))
    dA = _dev(lib, Ac); dB = _dev(lib, Bc); dD = _dev(lib, d)
    ldc = (N + 7) // 8 * 8
    dC = _empty(lib, M * ldc * (4 if c_f32 else 2))
    rc = lib.rsys_op_gemm_f8(dA, dB, dC, M, N, K, K, K, ldc, a_fmt, int(c_f32), dD, seg_cols, alt, kbs[0], kbs[1], kbs[2])
    assert rc == 0, L.last_error()
    raw = _get(lib, dC, (M, ldc), np.float32 if c_f32 else np.uint16)
    out = (raw if c_f32 else _bf16_val(raw))[:, :N]
    for p in (dA, dB, dC, dD):
        lib.rsys_dev_free(p)
    A64 = Av.astype(np.float64); B64 = Bv.astype(np.float64)
    if kb:
        ref = np.zeros((M, N))
        bounds = [0] + [b * 128 for b in kb] + [K]
        for j in range(len(kb) + 1):
            ks = slice(bounds[j], bounds[j + 1])
            ref += (A64[:, ks] @ B64[:, ks].T) * float(desc[j])
    else:
        col = np.arange(N)
        if alt:
            dc = np.where((col >> 4) & 1, desc[1], desc[0])
        elif seg_cols:
            dc = np.asarray(desc, np.float64)[col // seg_cols]
        else:
            dc = np.full(N, desc[0])
        ref = (A64 @ B64.T) * dc[None, :]
    return out, ref


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 512), (300, 264, 384), (1000, 200, 640), (2048, 1024, 1408),
                                   (9000, 2816, 256), (33000, 776, 256)])
@pytest.mark.parametrize("a_fmt", [0, 1])
def test_gemm_f8_exact(M, N, K, a_fmt):
    """values whose products and partial sums are exact: any wrong lane / k mapping of v_mfma_f32_16x16x128_f8f6f4 shows"""
    out, ref = run_gemm_f8(M, N, K, a_fmt=a_fmt, seed=M + N + K)
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    out, ref = run_gemm_f8(M, N, K, a_fmt=a_fmt, c_f32=False, desc=(0.5,), seed=M + N)
    np.testing.assert_array_equal(out, _bf16_val(_bf16_bits(ref.astype(np.float32))))


def test_gemm_f8_segment_descales():
    out, ref = run_gemm_f8(512, 1536, 512, desc=(0.5, 2.0, 0.25), seg_cols=512, seed=1)      # q | k | v weights
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    out, ref = run_gemm_f8(300, 2816, 512, desc=(0.5, 4.0), alt=1, seed=2)                    # [16 w1 | 16 w3] column blocks
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    out, ref = run_gemm_f8(600, 512, 1536, a_fmt=1, desc=(0.5, 2.0, 0.125), kb=(4, 8), seed=3)   # dq | dk | dv gradients (K segments)
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    out, ref = run_gemm_f8(600, 512, 1024, a_fmt=1, desc=(0.5, 2.0, 0.125), kb=(4, 6), seed=5)   # grouped-query heads: dq twice as wide
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    out, ref = run_gemm_f8(600, 512, 2816, a_fmt=1, desc=(2.0, 0.25), kb=(11,), seed=4)          # da | db
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    out, ref = run_gemm_f8(512, 1024, 512, desc=(0.5, 0.5, 2.0, 0.25), seg_cols=256, seed=6)     # q (two units) | k | v output columns
    np.testing.assert_array_equal(out, ref.astype(np.float32))


@pytest.mark.parametrize("a_fmt", [0, 1])
def test_gemm_f8_random_codes(a_fmt):
    """every finite code on both sides (magnitudes spread over 2^-9 .. 2^8): the sum inside one K = 128 instruction aligns its
    products to the largest one, so the smallest terms are lost -- an error of a few 1e-5 of the largest output, far below the
    fp8 rounding of the operands themselves (2^-4 relative)"""
    out, ref = run_gemm_f8(700, 520, 1408, a_fmt=a_fmt, desc=(3.1e-4,), small=False, seed=11)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    print("random codes: max error / max output", err)
    assert err < 3e-4, err


@pytest.mark.parametrize("a_fmt", [0, 1])
def test_gemm_f8_quantised_normal_data(a_fmt):
    """operands as the trunk produces them: normal data through the tensor-wise recipe"""
    from oracle import fp8
    lib, L = _lib()
    rng = np.random.default_rng(5 + a_fmt)
    M, N, K = 1024, 512, 1536
    a = rng.standard_normal((M, K)).astype(np.float32); b = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
    qa, sa = fp8.quantize(a, a_fmt); qb, sb = fp8.quantize(b, 0)
    d = np.zeros(32, np.float32); d[0] = fp8.descale(sa, sb)
    dA = _dev(lib, fp8.encode_fp8(qa, a_fmt)); dB = _dev(lib, fp8.encode_fp8(qb, 0)); dD = _dev(lib, d); dC = _empty(lib, M * N * 4)
    assert lib.rsys_op_gemm_f8(dA, dB, dC, M, N, K, K, K, N, a_fmt, 1, dD, 0, 0, 0, 0, 0) == 0, L.last_error()
    out = _get(lib, dC, (M, N), np.float32)
    for p in (dA, dB, dC, dD):
        lib.rsys_dev_free(p)
    ref = (qa.astype(np.float64) @ qb.astype(np.float64).T) * float(d[0])
    err = np.abs(out - ref).max() / np.abs(ref).max()
    exact = a.astype(np.float64) @ b.astype(np.float64).T
    qerr = np.abs(ref - exact).max() / np.abs(exact).max()
    print("quantised normal data: kernel vs exact sum of the fp8 products", err, "; fp8 quantisation itself", qerr)
    assert err < 6e-5, err     # (measured 1.6e-5 .. 1.8e-5: the instruction's internal alignment, as above)
    assert qerr < 0.1


@pytest.mark.parametrize("a_fmt", [0, 1])
def test_gemm_f8_against_torch_scaled_mm_on_the_gpu(a_fmt):
    """an independent checker on the device: PyTorch's own scaled fp8 matmul (hipBLASLt behind torch._scaled_mm) on the same bytes"""
    torch = pytest.importorskip("torch")
    if not (hasattr(torch, "_scaled_mm") and torch.cuda.is_available()):
        pytest.skip("no torch._scaled_mm on this device")
    from oracle import fp8
    lib, L = _lib()
    rng = np.random.default_rng(21 + a_fmt)
    M, N, K = 512, 256, 1024
    a = rng.standard_normal((M, K)).astype(np.float32); b = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    qa, sa = fp8.quantize(a, a_fmt); qb, sb = fp8.quantize(b, 0)
    ca, cb = fp8.encode_fp8(qa, a_fmt), fp8.encode_fp8(qb, 0)
    d = np.zeros(32, np.float32); d[0] = fp8.descale(sa, sb)
    dA = _dev(lib, ca); dB = _dev(lib, cb); dD = _dev(lib, d); dC = _empty(lib, M * N * 2)
    assert lib.rsys_op_gemm_f8(dA, dB, dC, M, N, K, K, K, N, a_fmt, 0, dD, 0, 0, 0, 0, 0) == 0, L.last_error()
    out = _bf16_val(_get(lib, dC, (M, N), np.uint16))
    for p in (dA, dB, dC, dD):
        lib.rsys_dev_free(p)
    try:
        ta = torch.from_numpy(ca).cuda().view(torch.float8_e4m3fn if a_fmt == 0 else torch.float8_e5m2)
        tb = torch.from_numpy(cb).cuda().view(torch.float8_e4m3fn)
        ia = torch.tensor(1.0 / float(sa), dtype=torch.float32, device="cuda"); ib = torch.tensor(1.0 / float(sb), dtype=torch.float32, device="cuda")
        want = torch._scaled_mm(ta, tb.t(), scale_a=ia, scale_b=ib, out_dtype=torch.bfloat16).float().cpu().numpy()
    except (RuntimeError, NotImplementedError) as e:
        pytest.skip(f"torch._scaled_mm not usable here: {e}")
    ulp = np.maximum(np.abs(want), 1e-30) * 2.0 ** -7
    assert (np.abs(out - want) <= ulp).all(), float(np.abs(out - want).max())      # one bf16 rounding of the output
    assert (out != want).mean() < 0.2
