"""Deterministic synthetic parameters, metadata tables and masks for the tests (numpy only).

TEST INFRASTRUCTURE.  Shared by the golden-fixture generator (oracle/gen_golden.py), the numpy oracle tests and the
CPU-baseline leg of bench.py.  Configurations and the synthetic corpus (batch records) live with the product's host
code (recommendersystem_amd/workload.py) and are re-exported here under their old names.
"""
import numpy as np

from recommendersystem_amd.workload import (F32_KEYS, INT_KEYS, MEDIUMS, METRICS3, PLANNED_STATUS, batch_keys,  # noqa: F401
                                            key_dtype, make_batch, make_config, make_stream)


def param_shapes(cfg):
    """Ordered {state_dict key: shape} of the reference model
    (transformer.model.py:27-49, 97-118, 216-234, 289-295, 312-333, 346-359).
    `watch_head.item_embedding.*` aliases are not listed (shared storage)."""
    D = cfg["embed_dim"]; I = cfg["intermediate_dim"]
    H = cfg["num_heads"]; KV = cfg["num_kv_heads"]; hd = D // H
    V = cfg["vocab_sizes"]["0_matchedid"] + cfg["vocab_sizes"]["1_matchedid"]
    M = cfg["metadata_emb_size"]
    vs = cfg["vocab_sizes"]
    sh = {}
    sh["action_embedding.periodic_time_cos"] = (2,)
    sh["action_embedding.periodic_time_sin"] = (2,)
    sh["action_embedding.status_embedding.embedding.weight"] = (vs["status"] + 1, 16)
    sh["action_embedding.gender_embedding.embedding.weight"] = (vs["gender"] + 1, 4)
    sh["action_embedding.source_embedding.embedding.weight"] = (vs["source"] + 1, 4)
    sh["action_embedding.linear.weight"] = (D, 32)
    sh["action_embedding.linear.bias"] = (D,)
    sh["item_embedding.matchedid_embedding.embedding.weight"] = (V + 1, D)
    sh["item_embedding.metadata_embedding.embedding.weight"] = (V + 1, M)
    sh["item_embedding.projection_layer.weight"] = (D, M)
    sh["item_embedding.projection_layer.bias"] = (D,)
    for l in range(cfg["num_layers"]):
        p = f"transformers.layers.{l}."
        sh[p + "attn.q_proj.weight"] = (H * hd, D)
        sh[p + "attn.k_proj.weight"] = (KV * hd, D)
        sh[p + "attn.v_proj.weight"] = (KV * hd, D)
        sh[p + "attn.output_proj.weight"] = (D, H * hd)
        if cfg.get("finetune"):
            sh[p + "attn.q_proj_lora_A.weight"] = (8, D)
            sh[p + "attn.q_proj_lora_B.weight"] = (H * hd, 8)
            sh[p + "attn.v_proj_lora_A.weight"] = (8, D)
            sh[p + "attn.v_proj_lora_B.weight"] = (KV * hd, 8)
        sh[p + "mlp.w1.weight"] = (I, D)
        sh[p + "mlp.w2.weight"] = (D, I)
        sh[p + "mlp.w3.weight"] = (I, D)
        sh[p + "sa_norm.scale"] = (D,)
        sh[p + "mlp_norm.scale"] = (D,)
    sh["transformers.norm.scale"] = (D,)
    sh["rating_head.0.weight"] = (D, D)
    sh["rating_head.0.bias"] = (D,)
    sh["rating_head.2.weight"] = (1, D)
    sh["rating_head.2.bias"] = (1,)
    return sh


FROZEN = ("item_embedding.metadata_embedding.embedding.weight",)


def trainable_names(cfg):
    names = [n for n in param_shapes(cfg) if n not in FROZEN]
    if cfg.get("finetune"):
        names = [n for n in names if "lora_" in n]  # transformer.model.py:361-371
    return names


def make_params(cfg, seed, style="test"):
    """style="init": the reference init (transformer.model.py:5-12: N(0,0.006),
    zero biases, last row of every embedding zeroed, norm scales 1, phases 0).
    style="test": every tensor random and O(1)-conditioned so that every
    gradient path is exercised (biases, norm scales, phases non-trivial)."""
    rng = np.random.default_rng(seed)
    P = {}
    D = cfg["embed_dim"]
    for name, shape in param_shapes(cfg).items():
        if style == "init":
            if name.endswith(".scale"):
                w = np.ones(shape, np.float32)
            elif name.endswith(".bias") or "periodic" in name:
                w = np.zeros(shape, np.float32)
            else:
                w = (rng.standard_normal(shape) * 0.006).astype(np.float32)
                if "embedding.weight" in name:
                    w[-1] = 0
                if "lora_B" in name:
                    w[:] = 0
        else:
            if name.endswith(".scale"):
                w = (1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32)
            elif "periodic" in name:
                w = (0.5 * rng.standard_normal(shape)).astype(np.float32)
            elif name.endswith(".bias"):
                w = (0.05 * rng.standard_normal(shape)).astype(np.float32)
            elif "metadata_embedding" in name:
                w = (rng.standard_normal(shape) / np.sqrt(shape[1])).astype(np.float32)
            elif "embedding.weight" in name:
                w = (0.5 * rng.standard_normal(shape)).astype(np.float32)
            else:
                fan_in = shape[-1]
                w = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        if "metadata_embedding" in name:
            w[-1] = 0  # transformer.model.py:386 (mask row of the frozen table is zero)
        P[name] = w
    return P


def make_metadata(cfg, seed):
    """Frozen metadata table (V, M) f32, N(0,1)/sqrt(M)."""
    V = cfg["vocab_sizes"]["0_matchedid"] + cfg["vocab_sizes"]["1_matchedid"]
    M = cfg["metadata_emb_size"]
    rng = np.random.default_rng(seed)
    out = np.empty((V, M), np.float32)
    step = 8192
    for i in range(0, V, step):
        n = min(step, V - i)
        out[i:i + n] = rng.standard_normal((n, M), dtype=np.float32) / np.float32(np.sqrt(M))
    return out


def make_masks(cfg, rows, seed):
    """Pretraining masks from a uniform draw (transformer.model.py:437-440)."""
    S = cfg["max_sequence_length"]
    rng = np.random.default_rng(seed)
    u = rng.random((rows, S)).astype(np.float32)
    r = np.float32(cfg["mask_rate"])
    watch = u < r
    rating = (u >= r) & (u < 2 * r)
    return watch, rating
