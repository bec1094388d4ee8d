"""CPU oracle: numpy restatement of the reference's model arithmetic.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this; the product path (recommendersystem_amd)
never does and fails loudly without its HIP library.

Pinned: tests/test_oracle_golden.py checks this file against fixtures produced
by running the reference's own transformer.model.py (oracle/gen_golden.py):
masked batch (bit-exact), embeddings, trunk output, 4 train losses / 8 eval
values, every named gradient, grad norm and three clip+AdamW steps.

Every function cites the reference lines it restates; `model.py` =
/root/reference/notebooks/Training/transformer.model.py.
Forward and hand-derived backward, any float dtype (float64 default = "truth";
float32 for the timed CPU baseline).
"""
import math

import numpy as np

ALL_MEDIUMS = (0, 1)
ALL_METRICS = ("watch", "rating")
TASKS = [(m, k) for m in ALL_MEDIUMS for k in ALL_METRICS]  # model.py:503-504 order


# --------------------------------------------------------------------------- masks
def mask_tokens(cfg, d, watch_mask=None, rating_mask=None):
    """model.py:417-462.  d: dict of (rows,S) arrays (copied).  Pretraining takes
    the two boolean masks (the reference draws them from torch.rand, :437-440);
    finetune derives them from the chosen metric's weights (:418-435)."""
    d = {k: np.array(v) for k, v in d.items()}
    if cfg.get("finetune"):
        wm = np.zeros(d["userid"].shape, bool)
        rm = np.zeros(d["userid"].shape, bool)
        metric = cfg["finetune_metric"]
        for k in d:
            if metric in k and k.endswith(".weight"):
                if metric == "watch":
                    wm |= d[k] > 0
                else:
                    rm |= d[k] > 0
        watch_mask, rating_mask = wm, rm
    watch_mask = np.asarray(watch_mask, bool)
    rating_mask = np.asarray(rating_mask, bool)
    d["token_mask_ids"] = d["token_mask_ids"] * rating_mask.astype(d["token_mask_ids"].dtype)
    both = watch_mask | rating_mask
    for k in d:
        if k.endswith(".position") or k.endswith(".label") or k.endswith(".weight"):
            if "watch" in k:
                d[k][~watch_mask] = 0
            elif "rating" in k:
                d[k][~rating_mask] = 0
        elif k == "matchedid":
            d[k][watch_mask] = -1
        elif k == "status":
            d[k][both] = -1
        elif k in ("rating", "progress"):
            d[k][both] = 0
    return d


def select_positions(w_flat, topk):
    """model.py:509 `torch.topk(w.reshape(-1), k=topk)` made deterministic:
    positive-weight positions in ascending flat index, then zero-weight positions
    in ascending flat index (zero-weight rows contribute 0 to every loss, so this
    equals the reference whenever #(w>0) <= topk; on overflow the reference's tie
    order is implementation-defined).  Positions are ordered by
    weight first (largest first) only when weights differ, as topk does."""
    w_flat = np.asarray(w_flat)
    order = np.lexsort((np.arange(w_flat.size), -w_flat.astype(np.float64)))
    return order[:topk].astype(np.int64)


# --------------------------------------------------------------------------- pieces
def rope_tables(head_dim, end, theta=500000.0):
    """model.py:173-179 in float32 like torch."""
    freqs = (1.0 / (np.float32(theta) ** (np.arange(0, head_dim, 2, dtype=np.float32)[: head_dim // 2] / np.float32(head_dim)))).astype(np.float32)
    t = np.arange(end, dtype=np.float32)
    f = np.outer(t, freqs).astype(np.float32)
    return np.cos(f).astype(np.float32), np.sin(f).astype(np.float32)


def apply_rope(x, cos, sin):
    """model.py:182-190; x (B,T,h,hd), cos/sin (B,T,hd/2) or (T,hd/2): interleaved pairs."""
    if cos.ndim == 2:
        cos = cos[None]; sin = sin[None]
    c = cos[:, :, None, :]; s = sin[:, :, None, :]
    x0 = x[..., 0::2]; x1 = x[..., 1::2]
    out = np.empty_like(x)
    out[..., 0::2] = x0 * c - x1 * s
    out[..., 1::2] = x0 * s + x1 * c
    return out


def apply_rope_bwd(g, cos, sin):
    if cos.ndim == 2:
        cos = cos[None]; sin = sin[None]
    c = cos[:, :, None, :]; s = sin[:, :, None, :]
    g0 = g[..., 0::2]; g1 = g[..., 1::2]
    out = np.empty_like(g)
    out[..., 0::2] = g0 * c + g1 * s
    out[..., 1::2] = -g0 * s + g1 * c
    return out


def rmsnorm(x, scale, eps=1e-5):
    """model.py:193-202."""
    r = 1.0 / np.sqrt((x * x).mean(-1, keepdims=True) + eps)
    return x * r * scale, r


def rmsnorm_bwd(g, x, scale, r):
    D = x.shape[-1]
    gs = g * scale
    dscale = (g * x * r).reshape(-1, D).sum(0)
    dx = r * gs - x * (r ** 3) * ((gs * x).sum(-1, keepdims=True) / D)
    return dx, dscale


def attention_mask(userid_t, tmid_t):
    """model.py:479-487: (B,T,T) bool, [b,q,kv] allowed iff same userid AND
    (tmid[kv]==0 OR tmid[q]==tmid[kv]); no causal term."""
    doc = userid_t[:, :, None] == userid_t[:, None, :]
    tok = (tmid_t[:, None, :] == 0) | (tmid_t[:, :, None] == tmid_t[:, None, :])
    return doc & tok


def interleave(x, y):
    """model.py:403-415."""
    return np.stack([x, y], axis=2).reshape(x.shape[0], x.shape[1] * 2, *x.shape[2:])


def gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def gelu_grad(x):
    from scipy.special import erf
    return 0.5 * (1.0 + erf(x / math.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def bf16_round(a):
    """float -> nearest-even bfloat16 -> float (the storage rounding of the HIP path's T-typed operands)."""
    a32 = np.ascontiguousarray(a, np.float32)
    u = a32.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(a32.shape)


def _remap(x, vocab):
    """model.py:23-24 MaskedEmbedding: -1 -> last row."""
    return np.where(x == -1, vocab, x)


# --------------------------------------------------------------------------- model
class OracleModel:
    """Forward/backward of RecommenderModel (model.py:346-538) on numpy arrays.
    P: {state_dict key: array}.  Gradients for the trainable keys only."""

    def __init__(self, cfg, P, dtype=np.float64, operand_round=None):
        """operand_round="bf16": every array the HIP path stores as a bf16 GEMM operand (weights' shadow copies,
        normalised activations, q/k/v, attention probabilities and outputs, SwiGLU products, logits, and the matching
        gradient operands of the backward) is rounded to bfloat16 at that point; accumulation stays in `dtype`.  This
        is the reference's autocast arithmetic as the benchmarked mode computes it, for the tight bf16 parity check.
        operand_round="fp8": the same, and the transformer blocks' linears (q k v o w1 w3 w2: forward and input-gradient
        products) take tensor-wise dynamically scaled fp8 operands -- torchao's "tensorwise" float8 recipe the reference applies
        to `transformers.*` for pretraining (transformer.py:671-676), restated in oracle/fp8.py (pinned to torch's float8 casts and torch._scaled_mm in tests/test_fp8_oracle.py; not against torchao
        itself): forward, input-gradient and weight-gradient products.  `fp8_dw = False` afterwards: weight gradients from the
        bf16 operands instead (the HIP path's RSYS_F8_DW=0 / deterministic mode)."""
        self.cfg = cfg
        self.dt = dtype
        self.fp8 = operand_round == "fp8"
        self.fp8_dw = self.fp8
        if operand_round is None:
            self.q = lambda a: a
        else:
            assert operand_round in ("bf16", "fp8")
            self.q = lambda a: bf16_round(a).astype(dtype)
        self.P = {k: np.asarray(v, dtype) for k, v in P.items()}
        D = cfg["embed_dim"]; H = cfg["num_heads"]
        self.hd = D // H
        self.cos, self.sin = rope_tables(self.hd, 2 * cfg["max_sequence_length"])
        self.V0 = cfg["vocab_sizes"]["0_matchedid"]
        self.V = self.V0 + cfg["vocab_sizes"]["1_matchedid"]
        self.lora = bool(cfg.get("finetune")) and any("lora_" in k for k in self.P)

    def W(self, name):
        """a weight as a GEMM operand (rounded like the bf16 shadow when operand rounding is on)"""
        return self.q(self.P[name])

    def lin(self, x, name):
        """y = x W^T of a trunk linear (torchao Float8Linear forward in fp8 mode: e4m3 input and weight, one scale per tensor)"""
        if not self.fp8:
            return x @ self.W(name).T
        from . import fp8
        qx, sx = fp8.quantize(np.asarray(x, np.float32), fp8.E4M3)
        qw, sw = fp8.quantize(np.asarray(self.P[name], np.float32), fp8.E4M3)
        return (qx.astype(self.dt) @ qw.astype(self.dt).T) * self.dt(fp8.descale(sx, sw))

    def lin_dx(self, g, name):
        """dx = g W (fp8 mode: e5m2 output gradient, e4m3 weight)"""
        if not self.fp8:
            return g @ self.W(name)
        from . import fp8
        qg, sg = fp8.quantize(np.asarray(g, np.float32), fp8.E5M2)
        qw, sw = fp8.quantize(np.asarray(self.P[name], np.float32), fp8.E4M3)
        return (qg.astype(self.dt) @ qw.astype(self.dt)) * self.dt(fp8.descale(sg, sw))

    def lin_dw(self, g, x):
        """dW = g^T x over all tokens (fp8 mode: e5m2 output gradient, e4m3 input, fp32 result -- torchao rounds this product to
        bf16 before it is accumulated into the fp32 .grad; the HIP path and this oracle keep the fp32 sum)"""
        g2 = g.reshape(-1, g.shape[-1]); x2 = x.reshape(-1, x.shape[-1])
        if not (self.fp8 and self.fp8_dw):
            return g2.T @ x2
        from . import fp8
        qg, sg = fp8.quantize(np.asarray(g2, np.float32), fp8.E5M2)
        qx, sx = fp8.quantize(np.asarray(x2, np.float32), fp8.E4M3)
        return (qg.astype(self.dt).T @ qx.astype(self.dt)) * self.dt(fp8.descale(sg, sx))

    # ---- embeddings
    def action_features(self, d):
        """model.py:51-93 -> (B,S,32) feature vector + pieces for backward."""
        cfg, P, dt = self.cfg, self.P, self.dt
        min_ts, max_ts = cfg["min_ts"], cfg["max_ts"]
        ts = np.maximum(d["time"].astype(np.float64), min_ts)          # :55 clip(min_ts) in f64
        per = np.stack([2 * np.pi * ts / 86400, 2 * np.pi * ts / 604800], -1)
        per = per.astype(np.float32)                                     # :62 cast to f32
        time_emb = ((ts - min_ts) / (max_ts - min_ts)).astype(np.float32).astype(dt)[..., None]  # :65
        # :67-68 the phase is added IN float32 (|per| ~ 1e5, ulp ~ 8e-3): part of the reference's arithmetic
        pc = (per + P["action_embedding.periodic_time_cos"].astype(np.float32)).astype(dt)
        ps = (per + P["action_embedding.periodic_time_sin"].astype(np.float32)).astype(dt)
        vs = cfg["vocab_sizes"]
        gi = _remap(d["gender"], vs["gender"]); si = _remap(d["source"], vs["source"])
        sti = _remap(d["status"], vs["status"])
        rating = d["rating"].astype(dt)
        has = (d["rating"] != 0).astype(dt)[..., None]                   # :73
        rat = has * ((rating[..., None] - dt(cfg["rating_mean"])) / dt(cfg["rating_std"]))  # :75-77
        feat = np.concatenate([
            time_emb, np.cos(pc), np.sin(ps),
            P["action_embedding.gender_embedding.embedding.weight"][gi],
            P["action_embedding.source_embedding.embedding.weight"][si],
            has, rat,
            P["action_embedding.status_embedding.embedding.weight"][sti],
            d["progress"].astype(dt)[..., None]], -1)                    # :80-93
        return feat, (pc, ps, gi, si, sti)

    def fused_table(self):
        """E + Meta Wp^T + bp over all V+1 rows (model.py:120-133 `fuse`;
        identical math to :143-145 per token and :165-169 per medium)."""
        P = self.P
        if "item_embedding.fused_embedding" in P:
            return P["item_embedding.fused_embedding"]
        return (P["item_embedding.matchedid_embedding.embedding.weight"]
                + self.W("item_embedding.metadata_embedding.embedding.weight") @ self.W("item_embedding.projection_layer.weight").T
                + P["item_embedding.projection_layer.bias"])

    # ---- trunk
    def trunk(self, x, mask, cos, sin, cache):
        cfg, P = self.cfg, self.P
        H, KV, hd = cfg["num_heads"], cfg["num_kv_heads"], self.hd
        B, T, D = x.shape
        rep = H // KV
        scale = 1.0 / math.sqrt(hd)
        neg = np.where(mask, 0.0, -np.inf).astype(self.dt)[:, None]     # (B,1,T,T)
        for l in range(cfg["num_layers"]):
            p = f"transformers.layers.{l}."
            c = {}
            Q = self.q
            xn, r1 = rmsnorm(x, P[p + "sa_norm.scale"])
            xn = Q(xn)
            q = self.lin(xn, p + "attn.q_proj.weight")
            k = self.lin(xn, p + "attn.k_proj.weight")
            v = self.lin(xn, p + "attn.v_proj.weight")
            if self.lora:                                               # model.py:263-271 (dropout = identity)
                c["qa"] = Q(xn @ self.W(p + "attn.q_proj_lora_A.weight").T)
                c["va"] = Q(xn @ self.W(p + "attn.v_proj_lora_A.weight").T)
                q = q + 2.0 * (c["qa"] @ self.W(p + "attn.q_proj_lora_B.weight").T)
                v = v + 2.0 * (c["va"] @ self.W(p + "attn.v_proj_lora_B.weight").T)
            q = Q(apply_rope(q.reshape(B, T, H, hd), cos, sin))
            k = Q(apply_rope(k.reshape(B, T, KV, hd), cos, sin))
            v = Q(v.reshape(B, T, KV, hd))
            kk = np.repeat(k, rep, axis=2); vv = np.repeat(v, rep, axis=2)   # GQA: head h -> kv head h//rep
            qh = q.transpose(0, 2, 1, 3); kh = kk.transpose(0, 2, 1, 3); vh = vv.transpose(0, 2, 1, 3)  # (B,H,T,hd)
            s = np.matmul(qh, kh.transpose(0, 1, 3, 2)) * scale + neg
            s = s - s.max(-1, keepdims=True)
            pr = np.exp(s); den = pr.sum(-1, keepdims=True)
            o = Q((np.matmul(Q(pr), vh) / den).transpose(0, 2, 1, 3).reshape(B, T, H * hd))   # (flash kernels round exp(s - max), not the quotient)
            pr /= den
            h = x + self.lin(o, p + "attn.output_proj.weight")
            hn, r2 = rmsnorm(h, P[p + "mlp_norm.scale"])
            hn = Q(hn)
            a = self.lin(hn, p + "mlp.w1.weight")
            b = self.lin(hn, p + "mlp.w3.weight")
            sig = 1.0 / (1.0 + np.exp(-a))
            g = Q(a * sig * b)
            a = Q(a); b = Q(b)                                          # (what the backward reads back; the product above used the accumulators)
            sig = 1.0 / (1.0 + np.exp(-a))
            out = h + self.lin(g, p + "mlp.w2.weight")
            c.update(x=x, xn=xn, r1=r1, q=q, k=k, v=v, pr=pr, o=o, h=h, hn=hn, r2=r2, a=a, b=b, sig=sig, g=g)
            cache.append(c)
            x = out
        y, rf = rmsnorm(x, P["transformers.norm.scale"])
        cache.append(dict(x=x, rf=rf))
        return self.q(y)

    def trunk_bwd(self, gy, cos, sin, cache, G):
        cfg, P = self.cfg, self.P
        H, KV, hd = cfg["num_heads"], cfg["num_kv_heads"], self.hd
        rep = H // KV
        scale = 1.0 / math.sqrt(hd)
        cf = cache[-1]
        gx, G["transformers.norm.scale"] = rmsnorm_bwd(gy, cf["x"], P["transformers.norm.scale"], cf["rf"])
        fl = lambda t: t.reshape(-1, t.shape[-1])
        for l in reversed(range(cfg["num_layers"])):
            p = f"transformers.layers.{l}."
            c = cache[l]
            B, T, D = c["x"].shape
            # out = h + g W2^T
            Q = self.q
            gxq = Q(gx)                                                 # the gradient as a GEMM operand
            G[p + "mlp.w2.weight"] = self.lin_dw(gxq, c["g"])
            gg = self.lin_dx(gxq, p + "mlp.w2.weight")
            ga = Q(gg * c["b"] * (c["sig"] * (1.0 + c["a"] * (1.0 - c["sig"]))))
            gb = Q(gg * c["a"] * c["sig"])
            G[p + "mlp.w1.weight"] = self.lin_dw(ga, c["hn"])
            G[p + "mlp.w3.weight"] = self.lin_dw(gb, c["hn"])
            ghn = Q(self.lin_dx(ga, p + "mlp.w1.weight") + self.lin_dx(gb, p + "mlp.w3.weight"))
            dh, G[p + "mlp_norm.scale"] = rmsnorm_bwd(ghn, c["h"], P[p + "mlp_norm.scale"], c["r2"])
            gh = gx + dh
            ghq = Q(gh)
            # h = x + o Wo^T
            G[p + "attn.output_proj.weight"] = self.lin_dw(ghq, c["o"])
            go = Q(self.lin_dx(ghq, p + "attn.output_proj.weight")).reshape(B, T, H, hd)
            kk = np.repeat(c["k"], rep, axis=2); vv = np.repeat(c["v"], rep, axis=2)
            pr = c["pr"]
            goh = go.transpose(0, 2, 1, 3); kh = kk.transpose(0, 2, 1, 3); vh = vv.transpose(0, 2, 1, 3)
            qh = c["q"].transpose(0, 2, 1, 3)
            prT = pr.transpose(0, 1, 3, 2)
            gvv = np.matmul(Q(prT), goh).transpose(0, 2, 1, 3)         # (the flash kernels feed P and dS' = P (dP - delta) to the MFMA
            gp = np.matmul(goh, vh.transpose(0, 1, 3, 2))             #  as bf16 operands; 1/sqrt(hd) is applied to the finished sums)
            delta = (go * c["o"].reshape(B, T, H, hd)).sum(-1).transpose(0, 2, 1)[..., None]   # = rowsum(dP * P); from the STORED output, as the kernels do
            gs = Q(pr * (gp - delta))
            gq = np.matmul(gs, kh).transpose(0, 2, 1, 3) * scale
            gkk = np.matmul(gs.transpose(0, 1, 3, 2), qh).transpose(0, 2, 1, 3) * scale
            gk = gkk.reshape(B, T, KV, rep, hd).sum(3)
            gv = gvv.reshape(B, T, KV, rep, hd).sum(3)
            gq = Q(apply_rope_bwd(gq, cos, sin).reshape(B, T, H * hd))
            gk = Q(apply_rope_bwd(gk, cos, sin).reshape(B, T, KV * hd))
            gv = Q(gv.reshape(B, T, KV * hd))
            gxn = self.lin_dx(gq, p + "attn.q_proj.weight") + self.lin_dx(gk, p + "attn.k_proj.weight") + self.lin_dx(gv, p + "attn.v_proj.weight")
            if self.lora:
                G[p + "attn.q_proj_lora_B.weight"] = 2.0 * fl(gq).T @ fl(c["qa"])
                G[p + "attn.v_proj_lora_B.weight"] = 2.0 * fl(gv).T @ fl(c["va"])
                gqa = Q(2.0 * gq @ self.W(p + "attn.q_proj_lora_B.weight"))
                gva = Q(2.0 * gv @ self.W(p + "attn.v_proj_lora_B.weight"))
                G[p + "attn.q_proj_lora_A.weight"] = fl(gqa).T @ fl(c["xn"])
                G[p + "attn.v_proj_lora_A.weight"] = fl(gva).T @ fl(c["xn"])
                gxn = gxn + gqa @ self.W(p + "attn.q_proj_lora_A.weight") + gva @ self.W(p + "attn.v_proj_lora_A.weight")
            gxn = Q(gxn)
            G[p + "attn.q_proj.weight"] = self.lin_dw(gq, c["xn"])
            G[p + "attn.k_proj.weight"] = self.lin_dw(gk, c["xn"])
            G[p + "attn.v_proj.weight"] = self.lin_dw(gv, c["xn"])
            dx, G[p + "sa_norm.scale"] = rmsnorm_bwd(gxn, c["x"], P[p + "sa_norm.scale"], c["r1"])
            gx = gh + dx
        return gx

    # ---- full passes
    def embed(self, dm):
        """model.py:464-491 to_embedding on an already-masked batch: returns the
        trunk output (B,2S,D) and a cache for backward."""
        cfg, P = self.cfg, self.P
        feat, fc = self.action_features(dm)
        feat = self.q(feat)
        e_a = feat @ self.W("action_embedding.linear.weight").T + P["action_embedding.linear.bias"]
        Ft = self.fused_table()
        ids = _remap(dm["matchedid"], self.V)
        e_i = Ft[ids]
        x0 = interleave(e_i, e_a)
        uid = interleave(dm["userid"], dm["userid"])
        tm = interleave(dm["token_mask_ids"], dm["token_mask_ids"])
        mask = attention_mask(uid, tm)
        if "rope_input_pos" in dm:                                      # model.py:470-476
            pos = interleave(2 * dm["rope_input_pos"], 2 * dm["rope_input_pos"] + 1)
            cos = self.cos[pos].astype(self.dt); sin = self.sin[pos].astype(self.dt)
        else:
            T = x0.shape[1]
            cos = self.cos[:T].astype(self.dt); sin = self.sin[:T].astype(self.dt)
        cache = []
        y = self.trunk(x0, mask, cos, sin, cache)
        ctx = dict(feat=feat, fc=fc, ids=ids, Ft=Ft, cos=cos, sin=sin, cache=cache, e_a=e_a, e_i=e_i, x0=x0)
        return y, ctx

    def forward(self, dm, evaluate=False, want_grad=False, task_w=None):
        """model.py:493-529 train_forward on an already-masked (rows,S) batch.
        Returns losses (4 floats, or rating entries as 3-lists when evaluate) and,
        with want_grad, gradients of sum_i task_w[i]*loss_i for the trainable keys."""
        cfg, P, dt = self.cfg, self.P, self.dt
        y, ctx = self.embed(dm)
        B, T, D = y.shape
        S = T // 2
        e0 = y[:, 0::2].reshape(-1, D); e1 = y[:, 1::2].reshape(-1, D)
        topk = cfg["mask_topk"] * B
        losses = []
        gy = np.zeros_like(y) if want_grad else None
        G = {}
        Ft = ctx["Ft"]
        gFt = np.zeros_like(Ft) if want_grad else None
        for ti, (medium, metric) in enumerate(TASKS):
            w = dm[f"{medium}.{metric}.weight"].reshape(-1).astype(dt)
            lab = dm[f"{medium}.{metric}.label"].reshape(-1).astype(dt)
            pos = dm[f"{medium}.{metric}.position"].reshape(-1)
            bp = select_positions(w, topk)
            weights = w[bp]; labels = lab[bp]; positions = pos[bp].astype(np.int64)
            tw = 0.0 if task_w is None else float(task_w[ti])
            w_sum = max(weights.sum(), 1e-8)
            if metric == "watch":
                embed = e0[bp]
                s, e = (0, self.V0) if medium == 0 else (self.V0, self.V)
                items = self.q(Ft[s:e])
                logits = self.q(embed @ items.T)
                mx = logits.max(-1, keepdims=True)
                lse = mx[:, 0] + np.log(np.exp(logits - mx).sum(-1))
                ce = lse - logits[np.arange(len(bp)), positions]
                losses.append(float((ce * labels * weights).sum() / w_sum))
                if want_grad and tw != 0.0:
                    coef = (tw * labels * weights / w_sum)[:, None]
                    gl = np.exp(logits - lse[:, None])
                    gl[np.arange(len(bp)), positions] -= 1.0
                    gl = self.q(gl * coef)
                    tmp = np.zeros((B * S, D), dt)
                    np.add.at(tmp, bp, gl @ items)
                    gy[:, 0::2] += tmp.reshape(B, S, D)
                    gFt[s:e] += gl.T @ embed
            else:
                embed = e1[bp]
                W0, b0 = P["rating_head.0.weight"], P["rating_head.0.bias"]
                W2, b2 = P["rating_head.2.weight"], P["rating_head.2.bias"]
                W0 = self.q(W0)
                z = embed @ W0.T + b0
                hact = self.q(gelu(z))
                z = self.q(z)
                preds = (hact @ W2.T + b2).reshape(-1)
                tgt = labels - dt(cfg["rating_mean"])
                if evaluate:                                            # model.py:395-401 moments
                    losses.append([float((np.square(sc * preds - tgt) * weights).sum() / w_sum) for sc in (1.0, 0.0, -1.0)])
                else:
                    losses.append(float((np.square(preds - tgt) * weights).sum() / w_sum))
                if want_grad and tw != 0.0:
                    gp = (tw * 2.0 * (preds - tgt) * weights / w_sum)[:, None]
                    G["rating_head.2.weight"] = G.get("rating_head.2.weight", 0) + gp.T @ hact
                    G["rating_head.2.bias"] = G.get("rating_head.2.bias", 0) + gp.sum(0)
                    gz = self.q((gp @ W2) * gelu_grad(z))
                    G["rating_head.0.weight"] = G.get("rating_head.0.weight", 0) + gz.T @ embed
                    G["rating_head.0.bias"] = G.get("rating_head.0.bias", 0) + gz.sum(0)
                    ge = gz @ W0
                    tmp = np.zeros((B * S, D), dt)
                    np.add.at(tmp, bp, ge)
                    gy[:, 1::2] += tmp.reshape(B, S, D)
        if not want_grad:
            return losses
        return losses, self._backward(dm, ctx, gy, gFt, G)

    def _backward(self, dm, ctx, gy, gFt, G):
        cfg, P, dt = self.cfg, self.P, self.dt
        gx0 = self.trunk_bwd(gy, ctx["cos"], ctx["sin"], ctx["cache"], G)
        B, T, D = gx0.shape
        g_item = gx0[:, 0::2].reshape(-1, D); g_act = gx0[:, 1::2].reshape(-1, D)
        if gFt is not None and "item_embedding.fused_embedding" not in P:
            np.add.at(gFt, ctx["ids"].reshape(-1), g_item)
            Meta = P["item_embedding.metadata_embedding.embedding.weight"]
            G["item_embedding.matchedid_embedding.embedding.weight"] = gFt
            G["item_embedding.projection_layer.weight"] = self.q(gFt).T @ self.q(Meta)
            G["item_embedding.projection_layer.bias"] = gFt.sum(0)
        feat = ctx["feat"].reshape(-1, 32)
        pc, ps, gi, si, sti = ctx["fc"]
        G["action_embedding.linear.weight"] = g_act.T @ feat
        G["action_embedding.linear.bias"] = g_act.sum(0)
        g_act = self.q(g_act)
        gf = g_act @ self.W("action_embedding.linear.weight")                 # (N,32)
        pc = pc.reshape(-1, 2); ps = ps.reshape(-1, 2)
        G["action_embedding.periodic_time_cos"] = (-np.sin(pc) * gf[:, 1:3]).sum(0)
        G["action_embedding.periodic_time_sin"] = (np.cos(ps) * gf[:, 3:5]).sum(0)
        for name, idx, sl in (("gender", gi, slice(5, 9)), ("source", si, slice(9, 13)), ("status", sti, slice(15, 31))):
            key = f"action_embedding.{name}_embedding.embedding.weight"
            g = np.zeros_like(P[key])
            np.add.at(g, idx.reshape(-1), gf[:, sl])
            G[key] = g
        for k in list(G):
            if isinstance(G[k], (int, float)):
                G[k] = np.zeros_like(P[k])
        for k in P:
            if k not in G and k not in ("item_embedding.metadata_embedding.embedding.weight", "item_embedding.fused_embedding"):
                G[k] = np.zeros_like(P[k])
        if cfg.get("finetune"):
            G = {k: v for k, v in G.items() if "lora_" in k}
        return G

    def inference(self, d, task):
        """model.py:531-538."""
        y, _ = self.embed(d)
        if task == "retrieval":
            return y
        P = self.P
        z = y @ P["rating_head.0.weight"].T + P["rating_head.0.bias"]
        return gelu(z) @ P["rating_head.2.weight"].T + P["rating_head.2.bias"]


def reshape_batch(cfg, d_flat):
    """model.py:494-496: every array to (-1, S)."""
    S = cfg["max_sequence_length"]
    return {k: np.asarray(v).reshape(-1, S) for k, v in d_flat.items()}
