"""CPU: the fp8 restatement the GPU parity tests check against (oracle/fp8.py, oracle/model_np.py operand_round="fp8").
torchao is absent from the image; what is pinned here are the PyTorch primitives its Float8Linear is made of: OCP
e4m3fn / e5m2 round-to-nearest-even against torch's own float8 casts (and the byte codes), and the three products of a linear against
torch._scaled_mm.  The amax -> scale formula is checked in its published form only."""
import numpy as np
import pytest

from oracle import fp8


@pytest.mark.parametrize("fmt,tdt", [(fp8.E4M3, "float8_e4m3fn"), (fp8.E5M2, "float8_e5m2")])
def test_rounding_and_codes_match_torch_casts(fmt, tdt):
    torch = pytest.importorskip("torch")
    dt = getattr(torch, tdt)
    rng = np.random.default_rng(0)
    fm = fp8.fmax(fmt)
    x = np.concatenate([rng.standard_normal(50000).astype(np.float32) * s for s in (1e-5, 1e-3, 0.1, 1, 30, 400, 20000)])
    x = np.clip(x, -fm, fm).astype(np.float32)
    grid = fp8.decode_fp8(np.arange(256, dtype=np.uint8), fmt)
    g = np.sort(grid[np.isfinite(grid) & (np.abs(grid) <= fm)])
    mids = ((g[1:].astype(np.float64) + g[:-1]) / 2).astype(np.float32)          # exact ties: half to even
    x = np.concatenate([x, g, mids, np.nextafter(mids, np.float32(np.inf)), np.nextafter(mids, np.float32(-np.inf))])
    t = torch.from_numpy(x).to(dt)
    want = t.to(torch.float32).numpy(); want_b = t.view(torch.uint8).numpy()
    q = fp8.round_fp8(x, fmt)
    np.testing.assert_array_equal(q, want)
    b = fp8.encode_fp8(q, fmt)
    nz = q != 0
    np.testing.assert_array_equal(b[nz], want_b[nz])
    np.testing.assert_array_equal(fp8.decode_fp8(b, fmt), q)


def test_scale_follows_the_published_formula():
    assert fp8.scale_of(2.0, fp8.E4M3) == np.float32(224.0)
    assert fp8.scale_of(0.0, fp8.E4M3) == np.float32(448.0 / 1e-12)               # amax clamped at 1e-12
    assert fp8.scale_of(3.5, fp8.E5M2) == np.float32(57344.0 / 3.5)
    q, s = fp8.quantize(np.array([0.5, -2.0, 1.0], np.float32), fp8.E4M3)
    assert s == np.float32(224.0) and q.tolist() == [112.0, -448.0, 224.0]          # the largest magnitude lands on FMAX
    x = np.array([1.0, 3.0, -7.0], np.float32)
    q, s = fp8.quantize(x, fp8.E5M2)
    assert np.abs(q / s - x).max() <= 0.125 * np.abs(x).max()                       # two mantissa bits


def test_oracle_fp8_mode_runs_and_differs_from_bf16_by_fp8_noise():
    from oracle import model_np, synth
    cfg = synth.make_config("tiny", mask_rate=0.25, mask_topk=6)
    P = synth.make_params(cfg, 3, "test")
    d = synth.make_batch(cfg, 2, 4)
    wm, rm = synth.make_masks(cfg, 2, 5)
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
    tw = [0.05, 0.2, 0.3, 0.25]
    l16, G16 = model_np.OracleModel(cfg, P, np.float64, operand_round="bf16").forward(dm, False, True, tw)
    o8 = model_np.OracleModel(cfg, P, np.float64, operand_round="fp8")
    l8, G8 = o8.forward(dm, False, True, tw)
    assert all(np.isfinite(v) for v in l8)
    rel = max(abs(a - b) / max(abs(b), 1.0) for a, b in zip(l8, l16))
    assert 0 < rel < 0.2, rel
    n = "transformers.layers.0.mlp.w2.weight"
    gap = np.abs(G8[n] - G16[n]).max() / np.abs(G16[n]).max()
    assert 1e-3 < gap < 0.8, gap
    o8b = model_np.OracleModel(cfg, P, np.float64, operand_round="fp8"); o8b.fp8_dw = False
    _, G8b = o8b.forward(dm, False, True, tw)
    assert not np.array_equal(G8b[n], G8[n])                                        # weight gradients from bf16 operands: another arithmetic
    assert np.abs(G8b[n] - G8[n]).max() / np.abs(G8[n]).max() < 0.5


def _bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def test_oracle_linears_match_torch_scaled_mm():
    """The three products of a float8 linear as the oracle computes them (oracle/model_np.py lin / lin_dx / lin_dw) against PyTorch's own
    primitives -- float8 casts and torch._scaled_mm with inverse scales and a bf16 output, the call torchao's Float8Linear makes --
    on the CPU.  What stays restated from torchao's source, not pinned: the two-line amax -> scale formula and which tensors get
    which format."""
    torch = pytest.importorskip("torch")
    if not hasattr(torch, "_scaled_mm"):
        pytest.skip("no torch._scaled_mm")
    from oracle import model_np
    rng = np.random.default_rng(3)
    T, Din, Dout = 96, 64, 48
    x = _bf16_round(rng.standard_normal((T, Din)).astype(np.float32) * 1.7)
    g = _bf16_round(rng.standard_normal((T, Dout)).astype(np.float32) * 3e-4)
    W = (rng.standard_normal((Dout, Din)) * 0.04).astype(np.float32)
    ora = model_np.OracleModel.__new__(model_np.OracleModel)            # only the linear helpers are exercised
    ora.fp8 = True; ora.fp8_dw = True; ora.dt = np.float64; ora.P = {"w": W.astype(np.float64)}; ora.q = lambda a: a

    def q8(t, fmt):
        dt = torch.float8_e4m3fn if fmt == fp8.E4M3 else torch.float8_e5m2
        s = torch.tensor(fp8.scale_of(float(np.abs(t).max()), fmt))
        return (torch.from_numpy(t) * s).clamp(-fp8.fmax(fmt), fp8.fmax(fmt)).to(dt), (1.0 / s).float()

    def colmajor(t):
        return t.t().contiguous().t()

    try:
        x8, ix = q8(x, fp8.E4M3); w8, iw = q8(W, fp8.E4M3); g8, ig = q8(g, fp8.E5M2)
        y = torch._scaled_mm(x8, colmajor(w8.t()), scale_a=ix, scale_b=iw, out_dtype=torch.bfloat16).float().numpy()        # x W^T
        dx = torch._scaled_mm(g8, colmajor(w8), scale_a=ig, scale_b=iw, out_dtype=torch.bfloat16).float().numpy()           # g W
        dw = torch._scaled_mm(g8.t().contiguous(), colmajor(x8), scale_a=ig, scale_b=ix, out_dtype=torch.float32).numpy()   # g^T x
    except (RuntimeError, NotImplementedError) as e:                      # a build without the CPU kernel
        pytest.skip(f"torch._scaled_mm unavailable on this CPU build: {e}")

    def close_bf16(a, b):      # equal up to one bf16 rounding of the output (accumulation order at a rounding boundary)
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        ulp = np.maximum(np.abs(b), 1e-30) * 2.0 ** -7
        return (np.abs(a - b) <= ulp).all() and (a != b).mean() < 0.1

    assert close_bf16(_bf16_round(ora.lin(x.astype(np.float64), "w")), y)
    assert close_bf16(_bf16_round(ora.lin_dx(g.astype(np.float64), "w")), dx)
    ref_dw = ora.lin_dw(g.astype(np.float64), x.astype(np.float64))
    assert np.abs(ref_dw - dw).max() <= 2e-6 * np.abs(ref_dw).max()
