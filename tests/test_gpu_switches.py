"""The shape-of-work changes of round 3 change WHAT is launched, not what is computed: with each of them switched off
(RSYS_SPARSE_TOP=0: dense last layer and final norm; RSYS_TOP_ORDER=0: token order kept in the last layer's attention;
RSYS_DW_GROUP=0: one weight-gradient launch per product) a training step gives the same losses, the same dense trunk output, the
same gradients and the same updated parameters up to the order of float sums.  The switches are read once per process, so every
arm is its own process; this also keeps the switched-off code paths (the A/B arms of profiles/r3_*) exercised."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARMS = {"default": {}, "dense_top": {"RSYS_SPARSE_TOP": "0"}, "token_order": {"RSYS_TOP_ORDER": "0"}, "per_layer_dw": {"RSYS_DW_GROUP": "0"},
        "all_off": {"RSYS_SPARSE_TOP": "0", "RSYS_DW_GROUP": "0"},
        # deterministic mode (config["deterministic"], set by the worker): the grouped launch in its ordered form (per-product slabs,
        # one batched in-order sum; round 4) and the per-layer slab path it replaced
        "det_grouped_dw": {"RSYS_TEST_DETERMINISTIC": "1"}, "det_per_layer_dw": {"RSYS_TEST_DETERMINISTIC": "1", "RSYS_DET_DW_GROUP": "0"},
        # round 5 (csrc/switches.hpp): every kernel-choice switch whose other arm is live code has an arm here or a test of its own
        "select_one_pass": {"RSYS_SELECT_CHUNKED": "0"}, "select_aside": {"RSYS_SELECT_ASIDE": "1"},
        "scatter_atomic": {"RSYS_SCATTER_ATOMIC": "1"}, "dkdv_register_staged": {"RSYS_ATTN_KV_DMA": "0"},
        "side_stream_joined": {"RSYS_SIDE_STREAM": "1"}, "side_stream_deferred": {"RSYS_SIDE_STREAM": "2"},
        # round 6: the opt-in 128-query forward attention kernel (also with the compact top's q_active limit) and gemm8c's half tiles forced
        "fwd32": {"RSYS_ATTN_FWD32": "1"}, "gemm8c_half": {"RSYS_GEMM8C_HALF": "2", "RSYS_GEMM_KERNEL": "2"},
        # round 6, late: the four-wave register-named loops forced on (row-major plain stores, every eligible shape) / off (weight gradients back on the
        # eight-wave kernels), the mixed-layout dEw kernel forced, the reverse walk of the tile rows on every gemm8c launch
        "gemm4p_forced": {"RSYS_GEMM4P": "2", "RSYS_GEMM_KERNEL": "2"}, "gemm4k_off": {"RSYS_GEMM4K": "0", "RSYS_GEMM_KERNEL_TN": "2"},
        "gemm4k_forced_tn": {"RSYS_GEMM_KERNEL_TN": "2"}, "gemm_mix_forced": {"RSYS_GEMM_KERNEL_MIX": "2"},
        "reverse_everywhere": {"RSYS_GEMM_REVERSE": "2", "RSYS_GEMM_KERNEL": "2"}}


@pytest.mark.parametrize("dtype,tol_loss,tol", [("fp32", 1e-6, 2e-5), ("bf16", 2e-3, 3e-2)])
def test_switched_off_paths_compute_the_same_step(tmp_path, dtype, tol_loss, tol):
    res = {}
    for arm, env in ARMS.items():
        out = str(tmp_path / f"{arm}.npz")
        e = dict(os.environ, **env)
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_switch_worker.py"), out, dtype, ROOT], check=True, env=e, cwd=ROOT, timeout=300)
        res[arm] = np.load(out)
    ref = res["all_off"]
    for arm in ARMS:
        if arm == "all_off":
            continue
        z = res[arm]
        assert np.allclose(z["losses"], ref["losses"], rtol=tol_loss, atol=1e-7), (arm, z["losses"], ref["losses"])
        assert np.abs(z["trunk"] - ref["trunk"]).max() <= tol * max(np.abs(ref["trunk"]).max(), 1e-6), arm
        lr, flips, total = 1e-4, 0, 0
        for k in ref.files:
            if k[:2] == "g/":
                scale = max(np.abs(ref[k]).max(), 1e-12)
                assert np.abs(z[k] - ref[k]).max() <= tol * scale, (arm, k, float(np.abs(z[k] - ref[k]).max() / scale))
            elif k[:2] == "p/":
                # the first Adam step moves an element by lr g / (|g| + eps): elements whose gradient is within the summation noise of
                # zero may move the other way (2 lr apart), everything else lands on the same value
                diff = np.abs(z[k] - ref[k])
                assert diff.max() <= 2.04 * lr + 1e-7, (arm, k, float(diff.max()))
                flips += int((diff > 0.5 * lr).sum()); total += diff.size
        assert flips <= (2e-4 if dtype == "fp32" else 2e-2) * total, (arm, flips, total)
