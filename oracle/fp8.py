"""TEST INFRASTRUCTURE (oracle): OCP fp8 rounding and torchao's tensor-wise dynamic scaling, restated in numpy.

The reference converts the transformer blocks' linears with torchao's "tensorwise" float8 recipe
(/root/reference/notebooks/Training/transformer.py:671-676).  torchao is absent from this image, so this file restates the
recipe it publishes (torchao.float8: `amax_to_scale`, `hp_tensor_to_float8_dynamic`, `Float8LinearConfig` defaults: input and
weight e4m3, grad_output e5m2, one scale per tensor computed from this step's amax).  Pinned in tests/test_fp8_oracle.py to the
PyTorch primitives torchao's Float8Linear is made of, on the CPU: the rounding functions against torch's float8 casts (every grid
point and tie), the three products of a linear (oracle/model_np.py lin / lin_dx / lin_dw) against torch._scaled_mm with inverse scales.
NOT pinned (torchao itself is absent): the two-line amax -> scale formula and the format assignment, restated from its source.

    scale = float32(float64(FMAX) / max(float64(amax), 1e-12))        FMAX = 448 (e4m3fn), 57344 (e5m2)
    q(x)  = round-to-nearest-even(clamp(float32(x) * scale, -FMAX, FMAX)) on the fp8 grid
    y     = (sum_k q(a)[m,k] q(b)[n,k]) * (1/scale_a * 1/scale_b)      fp32 accumulation, output rounded to bf16
"""
import numpy as np

E4M3, E5M2 = 0, 1
_FMT = {E4M3: dict(mbits=3, emin=-6, bias=7, fmax=448.0), E5M2: dict(mbits=2, emin=-14, bias=15, fmax=57344.0)}


def fmax(fmt):
    return _FMT[fmt]["fmax"]


def scale_of(amax, fmt):
    """torchao.float8.float8_utils.amax_to_scale"""
    a = np.maximum(np.asarray(amax, np.float64), 1e-12)
    return (np.float64(fmax(fmt)) / a).astype(np.float32)


def round_fp8(v, fmt):
    """values of the fp8 grid nearest to v (ties to even), saturating; v float32 array, result float32"""
    f = _FMT[fmt]
    v = np.clip(np.asarray(v, np.float32).astype(np.float64), -f["fmax"], f["fmax"])
    a = np.abs(v)
    _, ex = np.frexp(a)                       # a = m 2^ex, m in [0.5, 1)
    e = np.maximum(ex - 1, f["emin"])         # exponent of the binade (subnormals share emin)
    quantum = np.ldexp(1.0, e - f["mbits"])
    q = np.rint(a / quantum) * quantum        # np.rint: half to even; a / quantum is exact
    return (np.sign(v) * np.minimum(q, f["fmax"])).astype(np.float32)


def encode_fp8(q, fmt):
    """byte codes of values that lie on the fp8 grid"""
    f = _FMT[fmt]
    q = np.asarray(q, np.float64)
    a = np.abs(q)
    _, ex = np.frexp(a)
    e = ex - 1
    sub = (a == 0) | (e < f["emin"])
    mant_sub = np.rint(a / np.ldexp(1.0, f["emin"] - f["mbits"])).astype(np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        mant_norm = np.rint((a / np.ldexp(1.0, np.where(sub, 0, e)) - 1.0) * (1 << f["mbits"])).astype(np.int64)
    code = np.where(sub, mant_sub, ((e + f["bias"]) << f["mbits"]) | mant_norm)
    return (code | (np.signbit(q).astype(np.int64) << 7)).astype(np.uint8)


def decode_fp8(b, fmt):
    f = _FMT[fmt]
    b = np.asarray(b, np.uint8).astype(np.int64)
    s = np.where(b & 0x80, -1.0, 1.0)
    ef = (b & 0x7F) >> f["mbits"]
    mant = b & ((1 << f["mbits"]) - 1)
    val = np.where(ef == 0, mant * np.ldexp(1.0, f["emin"] - f["mbits"]), (1.0 + mant / (1 << f["mbits"])) * np.ldexp(1.0, ef - f["bias"]))
    return (s * val).astype(np.float32)


def quantize(x, fmt, amax=None):
    """(values on the fp8 grid = q(x * scale), scale) for one tensor"""
    x = np.asarray(x, np.float32)
    if amax is None:
        amax = np.abs(x).max() if x.size else 0.0
    s = scale_of(amax, fmt)
    return round_fp8(x * s, fmt), s


def descale(sa, sb):
    return (np.float32(1.0) / np.float32(sa)) * (np.float32(1.0) / np.float32(sb))
