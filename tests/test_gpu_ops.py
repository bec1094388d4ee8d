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


def test_gemm_bf16_output():
    out, ref = run_gemm(1, 128, 256, 128, False, False, c_f32=False, seed=3)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 1e-2, err
