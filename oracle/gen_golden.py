"""Generate tests/golden/*.npz by RUNNING the reference's own model code on CPU.

TEST INFRASTRUCTURE, development container only.  The reference
(/root/reference) is read and exec'd at generation time; nothing of its text is
written to the repo -- only inputs-by-seed and numeric outputs (fixtures).

  python oracle/gen_golden.py            # rewrites tests/golden/*.npz

How the reference is driven: notebooks/Training/transformer.model.py
has no imports of its own (transformer.py:33-34 exec's it into its globals), so
it is exec'd here into a namespace that supplies torch/nn/F/np plus
CPU-defaulting wrappers for create_block_mask / flex_attention (the source file
itself is untouched).  Pure helper functions of transformer.py (WSDScheduler,
minimize_quadratic, make_task_weights, get_index_permutation, EarlyStopper) are
extracted by name with `ast` from the file (the module itself cannot be imported:
h5py/torchao are absent and it parses argv at import) and exec'd likewise.
"""
import ast
import functools
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.attention.flex_attention import and_masks, create_block_mask, flex_attention

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import synth  # noqa: E402

REF = "/root/reference/notebooks/Training"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def load_reference_model_ns():
    def flex_cpu(q, k, v, block_mask=None, enable_gqa=False, kernel_options=None):
        return flex_attention(q, k, v, block_mask=block_mask, enable_gqa=enable_gqa)

    ns = {"np": np, "torch": torch, "nn": nn, "F": F, "and_masks": and_masks,
          "create_block_mask": functools.partial(create_block_mask, device="cpu"),
          "flex_attention": flex_cpu}
    with open(f"{REF}/transformer.model.py") as f:
        exec(compile(f.read(), "transformer.model.py", "exec"), ns)
    return ns


def load_reference_train_fns(names, extra_ns):
    with open(f"{REF}/transformer.py") as f:
        src = f.read()
    tree = ast.parse(src)
    ns = dict(extra_ns)
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), "transformer.py", "exec"), ns)
    # PretrainDataset.get_index_permutation is a method: pull it out too
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "PretrainDataset":
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == "get_index_permutation":
                    exec(compile(ast.Module([sub], []), "transformer.py", "exec"), ns)
    return ns


def to_torch_batch(d, S):
    return {k: torch.from_numpy(np.array(v)).reshape(-1) for k, v in d.items()}


def state_dict_from(P):
    sd = {k: torch.from_numpy(v.copy()) for k, v in P.items()}
    for k in list(sd):
        if k.startswith("item_embedding."):
            sd["watch_head." + k] = sd[k]
    return sd


def summarize(a):
    a = np.asarray(a, np.float64).reshape(-1)
    idx = np.linspace(0, a.size - 1, num=min(64, a.size)).astype(np.int64)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum()), np.abs(a).max()], a[:32] if a.size >= 32 else np.pad(a, (0, 32 - a.size)), a[idx] if idx.size == 64 else np.pad(a[idx], (0, 64 - idx.size))])


def run_case(ns, name, cfg, rows, seed, full):
    S = cfg["max_sequence_length"]
    P = synth.make_params(cfg, seed, "test")
    d_np = synth.make_batch(cfg, rows, seed + 1)
    rng = np.random.default_rng(seed + 2)
    u = rng.random((rows, S)).astype(np.float32)
    task_w = np.array([0.05, 0.2, 0.3, 0.25], np.float64)

    torch.manual_seed(0)
    model = ns["RecommenderModel"](cfg)
    missing = model.load_state_dict(state_dict_from(P), strict=True)
    model.train()
    out = {}

    real_rand = torch.rand
    def fake_rand(shape, device=None):
        assert tuple(shape) == (rows, S)
        return torch.from_numpy(u.copy())

    def fwd(evaluate):
        d = to_torch_batch(d_np, S)
        torch.rand = fake_rand
        try:
            return model(d, evaluate), d
        finally:
            torch.rand = real_rand

    # masked batch + trunk output (transformer.model.py:497-498)
    d = to_torch_batch(d_np, S)
    for k in d:
        d[k] = d[k].reshape(-1, S)
    torch.rand = fake_rand
    try:
        dm = model.mask_tokens(d)
    finally:
        torch.rand = real_rand
    for k, v in dm.items():
        out["masked/" + k] = v.numpy().copy()
    with torch.no_grad():
        x_act = model.action_embedding(dm)
        x_item = model.item_embedding(dm["matchedid"])
        emb = model.to_embedding(dm)
    if full:
        out["act/action_embedding"] = x_act.numpy()
        out["act/item_embedding"] = x_item.numpy()
        out["act/trunk_out"] = emb.numpy()
    else:
        out["sum/action_embedding"] = summarize(x_act.numpy())
        out["sum/item_embedding"] = summarize(x_item.numpy())
        out["sum/trunk_out"] = summarize(emb.numpy())

    # losses, evaluate=False and True (transformer.model.py:493-529)
    losses, _ = fwd(False)
    out["loss/train"] = np.array([float(x) for x in losses], np.float64)
    ev, _ = fwd(True)
    flat = []
    for x in ev:
        flat += [float(y) for y in x] if isinstance(x, list) else [float(x)]
    out["loss/eval"] = np.array(flat, np.float64)

    # gradients of sum_i task_w[i]*loss[i] (transformer.py:264-272)
    model.zero_grad(set_to_none=True)
    losses, _ = fwd(False)
    total = sum(l * float(w) for l, w in zip(losses, task_w))
    total.backward()
    grads = {}
    for n, p in model.named_parameters():
        if p.requires_grad:
            g = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
            grads[n] = g.copy()
            out[("grad/" if full else "gsum/") + n] = g.copy() if full else summarize(g)
    gn = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    out["grad_norm"] = np.array([gn])

    # clip + AdamW, 3 steps on the same batch (transformer.py:273-276, 285-298; non-fused CPU AdamW)
    lr = 3e-3
    decay = [p for n, p in model.named_parameters() if p.requires_grad and p.dim() >= 2]
    nodecay = [p for n, p in model.named_parameters() if p.requires_grad and p.dim() < 2]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.1},
                             {"params": nodecay, "weight_decay": 0.0}], lr=lr, betas=(0.9, 0.95))
    step_losses = []
    norms = []
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        losses, _ = fwd(False)
        total = sum(l * float(w) for l, w in zip(losses, task_w))
        total.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)))
        opt.step()
        step_losses.append([float(x) for x in losses])
        if full and step == 1:
            # checkpoint after two steps in the reference's own format (transformer.py:456-466): the third step above is
            # then the known answer for "convert, resume, step" (tests: checkpoint interchange)
            fns = load_reference_train_fns(["WSDScheduler"], {"np": np})
            sched_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
            sched = torch.optim.lr_scheduler.LambdaLR(sched_opt, fns["WSDScheduler"](4, 40, 0.1, 0.1))
            sched_opt.step(); sched.step(); sched_opt.step(); sched.step()
            ckpt = {"model": {k: v.detach().clone() for k, v in model.state_dict().items()},
                    "optimizer": opt.state_dict(), "scheduler": sched.state_dict(), "config": dict(cfg), "epoch": 1,
                    "training_loss": [float(x) for x in losses], "test_loss": [float(x) for x in losses]}
            torch.save(ckpt, os.path.join(OUT, f"checkpoint_{name}.pt"))
            from recommendersystem_amd import checkpoint as _ck   # the converted form, so GPU tests need no torch
            np.savez_compressed(os.path.join(OUT, f"checkpoint_{name}_converted.npz"), **_ck.from_reference(ckpt))
    out["opt/lr"] = np.array([lr])
    out["opt/losses"] = np.array(step_losses, np.float64)
    out["opt/norms"] = np.array(norms, np.float64)
    for n, p in model.named_parameters():
        if p.requires_grad:
            v = p.detach().numpy()
            out[("opt/param/" if full else "opt/psum/") + n] = v.copy() if full else summarize(v)

    out["meta/u"] = u
    out["meta/task_w"] = task_w
    out["meta/seed"] = np.array([seed]); out["meta/rows"] = np.array([rows])
    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **out)
    print(name, "train losses", out["loss/train"], "grad_norm", gn)


def run_inference_case(ns, name, cfg, seed):
    """Inference forward with rope_input_pos / per-candidate token_mask_ids and the
    fused table (transformer.model.py:104-109,120-133,139-142,470-476,531-538)."""
    S = cfg["max_sequence_length"]
    cfg = dict(cfg); cfg["forward"] = "inference"; cfg["finetune"] = False
    P = synth.make_params(cfg, seed, "test")
    model = ns["RecommenderModel"](cfg)
    model.load_state_dict(state_dict_from(P), strict=True)
    model.eval()
    rows = 2
    d_np = synth.make_batch(cfg, rows, seed + 7)
    hist = S - 6
    d = {k: torch.from_numpy(np.array(v)).reshape(rows, S) for k, v in d_np.items()
         if "." not in k}
    d["userid"][:] = torch.arange(1, rows + 1, dtype=torch.int32)[:, None]
    pos = torch.arange(S, dtype=torch.int32)[None, :].repeat(rows, 1)
    tm = torch.zeros(rows, S, dtype=torch.int32)
    for j in range(hist, S):
        pos[:, j] = hist
        tm[:, j] = j - hist + 1
        d["status"][:, j] = -1; d["rating"][:, j] = 0; d["progress"][:, j] = 0
    d["token_mask_ids"] = tm
    d["rope_input_pos"] = pos
    with torch.no_grad():
        r = model(dict(d), "ranking").numpy()
        e = model(dict(d), "retrieval").numpy()
    out = {"in/" + k: v.numpy() for k, v in d.items()}
    out["out/ranking"] = r
    out["out/retrieval"] = e
    np.savez_compressed(os.path.join(OUT, f"infer_{name}.npz"), **out)
    print(name, "inference ranking head", r.reshape(rows, 2 * S)[:, -3:])


def run_finetune_case(ns, name, cfg, seed):
    """LoRA finetune forward/backward, dropout disabled via eval() on the dropout
    modules only (transformer.model.py:235-271,361-371,418-435)."""
    cfg = dict(cfg); cfg["finetune"] = True; cfg["finetune_metric"] = "rating"
    S = cfg["max_sequence_length"]
    rows = 3
    P = synth.make_params(cfg, seed, "test")
    model = ns["RecommenderModel"](cfg)
    sd = state_dict_from(P)
    sd = {k: v for k, v in sd.items()}
    model.load_state_dict(sd, strict=False)   # post hook fuses the item table
    model.train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.eval()
    d_np = synth.make_batch(cfg, rows, seed + 3)
    d = {k: torch.from_numpy(np.array(v)).reshape(rows, S) for k, v in d_np.items()}
    # finetune shards: one target per row = last rated event (F/transformer.jl:52-133)
    for k in list(d):
        if k.endswith(".weight") or k.endswith(".label") or k.endswith(".position"):
            if k.endswith(".position"):
                d[k] = d[k].to(torch.int64)   # transformer.py:182-183
    keep = torch.zeros(rows, S, dtype=torch.bool)
    for b in range(rows):
        w = d["1.rating.weight"][b] + d["0.rating.weight"][b]
        nz = torch.nonzero(w > 0).reshape(-1)
        if len(nz):
            keep[b, nz[-1]] = True
    for k in list(d):
        if k.endswith(".weight") or k.endswith(".label") or k.endswith(".position"):
            d[k] = d[k] * keep.to(d[k].dtype)
    out = {"in/" + k: v.numpy().copy() for k, v in d.items()}
    losses = model({k: v.clone() for k, v in d.items()}, False)
    out["loss/train"] = np.array([float(x) for x in losses])
    (losses[1] + 0.5 * losses[3]).backward()
    for n, p in model.named_parameters():
        if p.requires_grad:
            out["grad/" + n] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"finetune_{name}.npz"), **out)
    print(name, "finetune losses", out["loss/train"])


def run_host_fns():
    args = types.SimpleNamespace(finetune_metric=None, finetune_medium=None)
    ns = load_reference_train_fns(
        {"WSDScheduler", "minimize_quadratic", "make_task_weights", "EarlyStopper", "wsum"},
        {"np": np, "args": args, "ALL_MEDIUMS": [0, 1], "ALL_METRICS": ["watch", "rating"]})
    out = {}
    sc = ns["WSDScheduler"](warmup_steps=2000, total_steps=50000, decay_ratio=0.1, final_ratio=0.1)
    steps = np.array([0, 1, 1999, 2000, 2001, 44999, 45000, 45001, 47500, 49999, 50000, 60000])
    out["wsd/steps"] = steps
    out["wsd/factors"] = np.array([sc(int(s)) for s in steps], np.float64)
    sc2 = ns["WSDScheduler"](warmup_steps=10, total_steps=57, decay_ratio=0.1, final_ratio=0.1)
    out["wsd2/factors"] = np.array([sc2(s) for s in range(0, 60)], np.float64)
    out["task_w/pretrain"] = np.array(ns["make_task_weights"](), np.float64)
    for med in (0, 1):
        for met in ("watch", "rating"):
            args.finetune_medium, args.finetune_metric = med, met
            out[f"task_w/{med}.{met}"] = np.array(ns["make_task_weights"](), np.float64)
    ys = np.array([[3.0, 2.0, 5.0], [1.0, 1.0, 1.0], [0.7, 0.4, 0.9]])
    out["minq/y"] = ys
    out["minq/out"] = np.array([ns["minimize_quadratic"]([1, 0, -1], list(y)) for y in ys])
    # block shuffle (transformer.py:53-68) with a pinned numpy global seed
    arr = np.array([3, 3, 3, 7, 7, 1, 1, 1, 1, 9, 2, 2, 0, 0, 0], np.int32)
    np.random.seed(1234)
    perm = ns["get_index_permutation"](None, arr)
    np.random.seed(1234)
    out["shuffle/arr"] = arr
    out["shuffle/block_perm"] = np.random.permutation(int((arr[:-1] != arr[1:]).sum()) + 1)
    out["shuffle/index_perm"] = perm
    st = ns["EarlyStopper"](patience=2, rtol=0.001)
    scores = [1.0, 0.9, 0.8995, 0.8999, 0.85, 0.86, 0.87]
    rec = []
    for s in scores:
        st(s)
        rec.append([st.counter, float(st.early_stop), float(st.save_model)])
    out["stopper/scores"] = np.array(scores); out["stopper/rec"] = np.array(rec)
    np.savez_compressed(os.path.join(OUT, "host_fns.npz"), **out)
    print("host fns ok", out["task_w/pretrain"])


def run_serving_case():
    """Request -> batch -> response steps of the reference's embedding server (notebooks/Finetune/embed.py:27-161): its own
    `predict` runs on synthetic users with a recording stand-in for the model; the fixture holds the batch it built and what
    it extracted from a known `embs` tensor."""
    import contextlib
    with open("/root/reference/notebooks/Finetune/embed.py") as f:
        tree = ast.parse(f.read())
    ns = {"np": np, "torch": torch, "time": __import__("time"), "device": "cpu", "gpu_lock": contextlib.nullcontext()}
    ns["logger"] = types.SimpleNamespace(debug=lambda *a, **k: None)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("make_item", "tokenize", "project", "predict"):
            exec(compile(ast.Module([node], []), "embed.py", "exec"), ns)
    rng = np.random.default_rng(77)
    num_items_0 = 500

    def make_user(n_events, n_rank, gender):
        items, ts, hs, hr = [], 1.0e9, {}, {}
        mid = None
        for _ in range(n_events):
            if mid is None or rng.random() > 0.35:
                mid = (int(rng.integers(0, 2)), int(rng.integers(1, 400)))
            ts += float(rng.integers(1, 100000))
            st = int(rng.integers(0, 9)); rt = float(rng.integers(0, 11)) if rng.random() > 0.4 else 0.0
            if rng.random() < 0.25 and mid in hs:
                st, rt = hs[mid], hr[mid]                       # an event that changes nothing (dropped by project)
            items.append({"medium": mid[0], "matchedid": mid[1], "history_max_ts": ts, "status": st, "rating": rt,
                          "progress": float(rng.random()), "history_status": hs.get(mid, -1), "history_rating": hr.get(mid, 0.0)})
            hs[mid], hr[mid] = st, rt
        return {"user": {"gender": gender, "source": int(rng.integers(0, 4))}, "items": items, "timestamp": ts + 5.0,
                "ranking_items": [int(x) for x in rng.integers(1, 400, n_rank)]}

    users = [make_user(40, 5, None), make_user(3, 17, 1), make_user(2600, 9, 0)]   # the last one overflows 1023 history tokens
    out = {}
    import json
    out["users_json"] = np.frombuffer(json.dumps(users).encode(), np.uint8)
    for task, medium in (("retrieval", 1), ("ranking", 0)):
        rec = {}

        class Stub:
            module = types.SimpleNamespace(config={"vocab_sizes": {"0_matchedid": num_items_0}})

            def __call__(self, d, t):
                rec["d"] = {k: v.numpy().copy() for k, v in d.items()}
                n, L = d["userid"].shape
                width = 3 if t == "retrieval" else 1
                return (torch.arange(n * 2 * L * width, dtype=torch.float32) * 0.5).reshape(n, 2 * L, width)

        ns["models"] = {f"{medium}.{task}": Stub()}
        ret = ns["predict"](users, task, medium)
        for k, v in rec["d"].items():
            out[f"{task}/in/{k}"] = v
        out[f"{task}/ret_json"] = np.frombuffer(json.dumps(ret).encode(), np.uint8)
    out["num_items_0"] = np.array([num_items_0])
    np.savez_compressed(os.path.join(OUT, "serving.npz"), **out)
    print("serving fixture ok", {k: v.shape for k, v in out.items() if k.endswith("userid")})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ns = load_reference_model_ns()
    tiny = synth.make_config("tiny", mask_rate=0.25, mask_topk=6)
    run_case(ns, "tiny", tiny, rows=3, seed=11, full=True)
    hd64 = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    run_case(ns, "hd64", hd64, rows=2, seed=23, full=False)
    run_inference_case(ns, "tiny", tiny, seed=31)
    run_finetune_case(ns, "tiny", tiny, seed=41)
    run_host_fns()
    run_serving_case()


if __name__ == "__main__":
    main()
