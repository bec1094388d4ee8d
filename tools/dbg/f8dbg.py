import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import recommendersystem_amd as ra
from oracle import model_np, synth, fp8
TASK_W = [0.05, 0.2, 0.3, 0.25]
cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16)
rows, seed = 3, 31
P = synth.make_params(cfg, seed, "test")
d = synth.make_batch(cfg, rows, seed + 1)
wm, rm = synth.make_masks(cfg, rows, seed + 2)
dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
def rel(a,b):
    a=np.asarray(a,np.float64); b=np.asarray(b,np.float64)
    return (round(float(np.abs(a-b).max()/max(np.abs(b).max(),1e-30)),5), round(float(np.sqrt(((a-b)**2).sum()/max((b**2).sum(),1e-30))),5))
out={}
for mode in ("fp8","bf16"):
    ref = model_np.OracleModel(cfg, P, np.float64, operand_round=mode)
    cache=[]
    # run embed to get cache: use internal API
    y, c = ref.embed(dm)
    out[mode]=(y,c)
print(type(out["fp8"][1]), (out["fp8"][1].keys() if isinstance(out["fp8"][1],dict) else len(out["fp8"][1])))
for dt in ("fp8","bf16"):
    model = ra.RecommenderModel(cfg, dtype=dt, max_rows=rows)
    model.load_state_dict(P); model.set_loss_weights(TASK_W, 1)
    model(d, False, masks=(wm, rm))
    c = out[dt][1]
    tc = c["cache"]
    print(dt, "x0", rel(model.debug_get("embed.x0", rows), np.asarray(c["x0"]).reshape(-1, cfg["embed_dim"])) if isinstance(c,dict) and "x0" in c else "?")
    if tc is not None:
        for l in range(cfg["num_layers"]):
            cl = tc[l]
            D=cfg["embed_dim"]
            for f,key in (("x","x"),("xn","xn"),("O","o"),("h","h"),("hn","hn"),("g","g")):
                got = model.debug_get(f"act.{l}.{f}", rows)
                print(dt, l, f, rel(got, np.asarray(cl[key]).reshape(got.shape[0], -1)))
            q = np.asarray(cl["q"]).reshape(-1, cfg["num_heads"]*ref.hd); k = np.asarray(cl["k"]).reshape(q.shape[0], -1); v = np.asarray(cl["v"]).reshape(q.shape[0], -1)
            got = model.debug_get(f"act.{l}.qkv", rows)
            print(dt, l, "qkv", rel(got, np.concatenate([q,k,v],1)))
    print(dt, "trunk (max, rms)", rel(model.trunk_output(rows), out[dt][0]), "oracles apart", rel(out["bf16"][0], out["fp8"][0]))
    ref = model_np.OracleModel(cfg, P, np.float64, operand_round=dt)
    l_ref, G_ref = ref.forward(dm, False, True, TASK_W)
    _, G_other = model_np.OracleModel(cfg, P, np.float64, operand_round="bf16" if dt == "fp8" else "fp8").forward(dm, False, True, TASK_W)
    for n in synth.trainable_names(cfg):
        if "layers.0" in n or "layers" not in n:
            print("   grad", n, rel(model.grad(n), G_ref[n]), "oracles apart", rel(G_other[n], G_ref[n]))
    if dt=="fp8":
        print("aamax", model.debug_get("f8.aamax", rows).max(1)[0][:4], "wamax", model.debug_get("f8.wamax", rows)[0])
        xn = np.asarray(tc[0]["xn"]); print("oracle amax xn", np.abs(xn).max(), "O", np.abs(np.asarray(tc[0]["o"])).max(), "hn", np.abs(np.asarray(tc[0]["hn"])).max(), "g", np.abs(np.asarray(tc[0]["g"])).max())
        for nme in ("attn.q_proj","attn.k_proj","attn.v_proj","attn.output_proj","mlp.w1","mlp.w3","mlp.w2"):
            print(nme, np.abs(P[f"transformers.layers.0.{nme}.weight"]).max())
    model.close()
