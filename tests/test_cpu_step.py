"""The C++ / OpenMP restatement of the training step (oracle/cpu_step.cpp, bench.py's CPU baseline) against the numpy oracle
(oracle/model_np.py, pinned to the reference's own outputs by test_oracle_golden.py): SGEMM, 4 losses, every named gradient,
gradient norm and the parameters after clip + AdamW.  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import cpu_step, model_np, synth, train_np


def test_sgemm_matches_numpy_on_ragged_shapes():
    rng = np.random.default_rng(0)
    for M, N, K in ((1, 1, 1), (7, 17, 5), (97, 33, 300), (200, 515, 64), (48, 1024, 64), (301, 129, 513)):
        A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
        C = cpu_step.sgemm_nt(A, B)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        assert np.abs(C - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()) * np.sqrt(K), (M, N, K)


def test_avx2_kernel_when_forced():
    """the narrow micro kernel (hosts without AVX-512) in a fresh process: CPU_STEP_ISA is read when the library loads"""
    code = ("import numpy as np; from oracle import cpu_step as c; assert c.lib().cpu_step_isa() == 256; r = np.random.default_rng(1);"
            "A = r.standard_normal((77, 300)).astype(np.float32); B = r.standard_normal((45, 300)).astype(np.float32);"
            "assert np.abs(c.sgemm_nt(A, B) - A.astype(np.float64) @ B.astype(np.float64).T).max() < 1e-3")
    env = dict(os.environ, CPU_STEP_ISA="avx2", PYTHONPATH=ROOT)
    subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd=ROOT)


def _case(name, rows, seed, style, **over):
    cfg = synth.make_config(name, **over)
    P = {k: v.astype(np.float32) for k, v in synth.make_params(cfg, seed, style).items()}
    d = synth.make_batch(cfg, rows, seed + 1, mu=2.0, sigma=0.8)
    wm, rm = synth.make_masks(cfg, rows, seed + 2)
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
    return cfg, P, dm


@pytest.mark.parametrize("name,rows,style", [("tiny", 3, "test"), ("tiny", 4, "init"), ("hd64", 2, "test")])
def test_losses_and_gradients_match_the_numpy_oracle(name, rows, style):
    cfg, P, dm = _case(name, rows, 11, style)
    tw = train_np.make_task_weights()
    ref = model_np.OracleModel(cfg, P, np.float64)
    ref_losses, ref_G = ref.forward(dm, False, True, tw)
    cs = cpu_step.CpuStep(cfg, P)
    losses, G = cs.forward_backward(dm, tw)
    assert np.allclose(losses, ref_losses, rtol=2e-5, atol=1e-6), (losses, ref_losses)
    for k in synth.trainable_names(cfg):
        scale = max(np.abs(ref_G[k]).max(), 1e-12)
        assert np.abs(G[k] - ref_G[k]).max() <= 2e-4 * scale + 1e-9, (k, np.abs(G[k] - ref_G[k]).max(), scale)


@pytest.mark.parametrize("name,rows,tol,check_gap", [("tiny", 3, 3e-2, False), ("hd64", 2, 1e-2, True)])
def test_bf16_operand_rounding_matches_the_numpy_oracle(name, rows, tol, check_gap):
    """operand_round="bf16": the C++ step rounds at the numpy oracle's rounding points (what the bench-shape GPU parity test
    compares the benchmarked arithmetic with).  Roundings amplify last-bit differences of the float sums (a value on a rounding
    boundary flips a whole bf16 ulp: the numpy oracle run in float32 differs from itself in float64 by as much), hence looser bounds
    than the unrounded comparison; the forward's rounding points show in the watch losses (equal to 1e-6), the backward's in the
    gradients at hd64, where the two restatements are several times closer to each other than the rounded model is to the exact one."""
    cfg, P, dm = _case(name, rows, 11, "test")
    meta = "item_embedding.metadata_embedding.embedding.weight"
    P[meta] = model_np.bf16_round(P[meta])
    tw = train_np.make_task_weights()
    ref_losses, ref_G = model_np.OracleModel(cfg, P, np.float64, operand_round="bf16").forward(dm, False, True, tw)
    exact_losses, exact_G = model_np.OracleModel(cfg, P, np.float64).forward(dm, False, True, tw)
    losses, G = cpu_step.CpuStep(cfg, P, operand_round="bf16").forward_backward(dm, tw)
    assert np.allclose(losses, ref_losses, rtol=1e-3, atol=1e-6), (losses, ref_losses)     # (a rating loss over a few rows: one flipped ulp is 6e-4)
    assert not np.allclose(losses, exact_losses, rtol=1e-4, atol=1e-6)
    worst, gap = 0.0, 0.0
    for k in synth.trainable_names(cfg):
        scale = max(np.abs(ref_G[k]).max(), 1e-12)
        worst = max(worst, np.abs(G[k] - ref_G[k]).max() / scale)
        gap = max(gap, np.abs(exact_G[k] - ref_G[k]).max() / scale)
    assert worst <= tol, worst
    if check_gap:
        assert gap > 3 * worst, (gap, worst)      # the rounding mode moves the gradients more than the two restatements differ


def test_repeated_userid_in_one_row_takes_the_whole_row_path():
    """a userid that comes back later in the same row (never produced by the packer, allowed by the mask rule)"""
    cfg, P, dm = _case("tiny", 2, 5, "test")
    dm = {k: v.copy() for k, v in dm.items()}
    dm["userid"][0, :] = np.array([3, 3, 3, 9, 9, 3, 3, 3] * (cfg["max_sequence_length"] // 8))
    tw = train_np.make_task_weights()
    ref_losses, ref_G = model_np.OracleModel(cfg, P, np.float64).forward(dm, False, True, tw)
    losses, G = cpu_step.CpuStep(cfg, P).forward_backward(dm, tw)
    assert np.allclose(losses, ref_losses, rtol=2e-5, atol=1e-6)
    k = "transformers.layers.0.attn.k_proj.weight"
    assert np.abs(G[k] - ref_G[k]).max() <= 2e-4 * np.abs(ref_G[k]).max()


def test_clip_and_adamw_step_matches_the_numpy_oracle():
    cfg, P, dm = _case("tiny", 3, 23, "test")
    tw = train_np.make_task_weights()
    names = synth.trainable_names(cfg)
    ref = model_np.OracleModel(cfg, P, np.float64)
    _, ref_G = ref.forward(dm, False, True, tw)
    ref_G, ref_norm = train_np.clip_grad_norm({k: ref_G[k] for k in names}, 1.0)
    opt = train_np.AdamW(ref.P, names, 1e-3)
    P2 = dict(ref.P); opt.step(P2, ref_G)
    cs = cpu_step.CpuStep(cfg, P, lr=1e-3)
    cs.forward_backward(dm, tw)
    norm = cs.clip_adamw()
    assert abs(norm - ref_norm) <= 1e-4 * ref_norm
    for k in names:
        assert np.abs(cs.P[k] - P2[k]).max() <= 2e-5 * max(np.abs(P2[k]).max(), 1e-3), k


def test_bad_index_is_rejected():
    cfg, P, dm = _case("tiny", 2, 3, "test")
    dm = {k: v.copy() for k, v in dm.items()}
    dm["matchedid"][0, 0] = 10 ** 6
    with pytest.raises(ValueError):
        cpu_step.CpuStep(cfg, P).forward_backward(dm, train_np.make_task_weights())


def test_bench_cpu_baseline_leg_runs_on_a_small_configuration():
    """bench.py's cpu_baseline leg end to end (thread count from the cgroup quota, GEMM probe, two steps, buffers released)"""
    sys.path.insert(0, ROOT)
    import bench
    out = bench.cpu_baseline(synth.make_config("tiny"), 1, rows=4)
    assert out["kind"] == "port" and out["unit"] == "interactions/sec" and out["value"] > 0
    assert 1 <= out["cores"] <= (os.cpu_count() or 1) and out["cores"] == cpu_step.host_cpus()
    assert "4 rows" in out["sample"] and "no extrapolation" in out["sample"]
