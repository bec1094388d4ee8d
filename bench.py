"""Benchmark of the training hot path (BASELINE.json metric): interactions/sec of
forward + backward + gradient all-reduce + clip + AdamW on synthetic histories.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Workload = cfg-3 of SURVEY.md section 8 (retrieval d=512 seq=512, 200K items, L=8, 64 rows per GPU per
step): data-parallel, weak scaling (per-GPU rows fixed).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
MFMA_PEAK_TFLOPS_FP8 = 5000.0   # dense fp8 (v_mfma_f32_16x16x128_f8f6f4), same guide: the fp8 kernel forms ("8f", "8fs", "8gf") are priced against this
HBM_PEAK_GBS = 8000.0

# kernel behind a call-site tag: the library appends "@<kernel>" to every GEMM tag (gemm.hip: gemm_kernel_name)
KERNEL_SYMBOL = {   # substring of the kernel's name in rocprofv3 output
    "4p": "gemm4p_kernel",                                # plain bf16 stores with K >= 8192 on outputs >= 4 tiles wide: four waves of 128 x 128, the K loop one asm statement with named registers (gemm4p.hip)
    "8c": "gemm8c_kernel",                                # 256x256 LDS-DMA, persistent, ONE operand stream across a workgroup's output tiles (gemm8c.hip; one instantiation per epilogue class)
    "8p": "gemm8p_kernel<false, false>",                  # its predecessor: operand requests stop at the end of every output tile (gemm8p.hip; classes without an 8c kernel, RSYS_GEMM8C=0)
    "8s": "gemm8p_kernel<false, true>",                   # the same pipeline, row-major operands + split-K atomics
    "8t": "gemm8p_kernel<true, false>",                   # the same pipeline, K-major operands + split-K atomics
    "8ts": "gemm8p_kernel<true, false>",                  # the same kernel, one K split, fp32 output stored / accumulated (the tied head's table gradient)
    "4k": "gemm4k_kernel",                                # K-major operands + split-K atomics on four waves of 128 x 128 with a register-named asm loop (gemm4k.hip)
    "4kg": "gemm4k_group_kernel",                         # the same loop, all layers' weight gradients in one grouped launch
    "8g": "gemm8p_group_kernel",                          # the K-major pipeline, all layers' weight gradients in one grouped launch
    "8f": "gemm8p_f8_kernel",                             # --dtype fp8: the persistent pipeline on e4m3 / e5m2 operands (K tiles of 128)
    "8fs": "gemm8p_f8sk_kernel",                          # --dtype fp8: split-K weight gradients on transposed fp8 copies, one product per launch
    "8gf": "gemm8p_group_f8_kernel",                      # --dtype fp8: the same, all layers' products in one grouped launch
    "8m": "gemm8p_mix_kernel",                            # row-major A x K-major B, split-K atomics over device-side live rows (the tied head's dEw)
    "nt": "gemm_kernelIDF16bLb0ELb0ELb0ELb0E",      # 128x128 register-staged (gemm.hip)
    "nn": "gemm_kernelIDF16bLb0ELb0ELb0ELb1E",
    "tn": "gemm_kernelIDF16bLb0ELb0ELb1ELb1E",
}
KERNEL_LABEL = {"4p": "gemm4p_kernel (256x256 LDS-DMA, persistent, four waves of 128x128 with a register-named asm K loop, row-major bf16, plain store)",
                "8c": "gemm8c_kernel<epilogue class> (256x256 LDS-DMA, persistent, one operand stream per workgroup, row-major bf16)",
                "8p": "gemm8p_kernel<false, false> (256x256 LDS-DMA, persistent, row-major bf16)", "8s": "gemm8p_kernel<false, true> (256x256 LDS-DMA, row-major bf16, split-K)",
                "8t": "gemm8p_kernel<true, false> (256x256 LDS-DMA, K-major bf16, split-K)",
                "8ts": "gemm8p_kernel<true, false> (256x256 LDS-DMA, K-major bf16, one K split, plain fp32 store / accumulate)",
                "8g": "gemm8p_group_kernel (256x256 LDS-DMA, K-major bf16, split-K, grouped weight gradients of all layers)",
                "8f": "gemm8p_f8_kernel (256x256 LDS-DMA, persistent, row-major fp8 operands: e4m3 x e4m3 forward, e5m2 x e4m3 dx)",
                "8fs": "gemm8p_f8sk_kernel (256x256 LDS-DMA, fp8 e5m2 x e4m3 on K-contiguous copies, split-K weight gradient)",
                "8gf": "gemm8p_group_f8_kernel (the fp8 split-K form, grouped weight gradients of all layers)",
                "4k": "gemm4k_kernel (256x256 LDS-DMA, K-major bf16, split-K, four waves of 128x128 with a register-named asm loop)",
                "4kg": "gemm4k_group_kernel (the same loop, grouped weight gradients of all layers)",
                "8m": "gemm8p_mix_kernel (256x256 LDS-DMA, row-major A x K-major B, split-K atomics over the device-side live rows)",
                "nt": "gemm_kernel<bf16,NT>", "nn": "gemm_kernel<bf16,NN>", "tn": "gemm_kernel<bf16,TN>"}
def _latest_traffic_file():
    """the newest committed PMC summary (profiles/r<round><letter>_pmc_traffic.json, written by tools/prof_round.sh)"""
    import glob
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return names[-1] if names else "none"


TRAFFIC_FILE = _latest_traffic_file()


def traffic_of(db, variant):
    """HBM bytes per launch of the kernel from the committed PMC summary (profiles/<TRAFFIC_FILE>)."""
    sym = KERNEL_SYMBOL.get(variant)
    # a tag may stand for a family of instantiations (gemm8c_kernel<epilogue class>): launch-weighted mean over all of them
    hits = [k for name, k in db.items() if sym and sym in name]
    n = sum(k.get("launches", 1) for k in hits)
    return round(sum(k["hbm_bytes_per_launch"] * k.get("launches", 1) for k in hits) / n) if n else None


HBM_KERNEL_SYMBOL = {"hbm_gather": ["gather_items_kernel"], "hbm_scatter": ["seg_scatter_kernel", "seg_fixup_kernel"],   # (one call site, two launches)
                     "hbm_rmsnorm_fwd": ["rmsnorm_fwd_kernel"], "hbm_rmsnorm_bwd": ["rmsnorm_bwd_kernelIDF16b"], "adamw": ["adamw_kernel"],
                     "sumsq": ["sumsq_kernel"]}


def traffic_of_name(db, syms):
    """PMC HBM bytes per call site from the committed summary: for every symbol, the first kernel whose name contains it"""
    total = 0.0
    for sym in syms or []:
        hit = [k["hbm_bytes_per_launch"] for name, k in (db or {}).items() if sym in name]
        if not hit:
            return None
        total += hit[0]
    return total or None


def live_pmc_traffic(timeout_s=200):
    """HBM bytes per launch of every kernel of the default workload, measured NOW: two child `rocprofv3 --pmc` passes (FETCH_SIZE, then
    WRITE_SIZE: separate runs, no trace domains beside the counters -- MI355X_MICROARCH.md, HBM section) of this script on 4 steps, summarised with
    the guide's corrections by tools/pmc_traffic.py.  Children of this process, started after its timed region; returns (db, note)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    tmp = tempfile.mkdtemp(prefix="rsys_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    try:
        for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            cmd = [exe, "--pmc", counter, "-d", os.path.join(tmp, sub), "--output-format", "csv", "--", "python3", os.path.join(ROOT, "bench.py"),
                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-train-loop"]
            # (a session of its own: a pass that overruns is killed as a GROUP -- rocprofv3 AND the bench.py child it started, which
            # would otherwise go on using the GPU beside the legs measured next)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = proc.communicate(timeout=timeout_s)
            except BaseException:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                raise
            if proc.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {proc.returncode}): {err.decode(errors='replace')[-160:]}"
        fetch = pmc_traffic.collect(os.path.join(tmp, "fetch"), "FETCH_SIZE")
        write = pmc_traffic.collect(os.path.join(tmp, "write"), "WRITE_SIZE")
        db = {}
        for k in set(fetch) | set(write):
            fs, fn = fetch.get(k, (0.0, 0)); ws, wn = write.get(k, (0.0, 0))
            db[k] = {"launches": max(fn, wn, 1), "hbm_bytes_per_launch": (2 * fs / max(fn, 1) + ws / max(wn, 1)) * 1024}
        if not db:
            return None, "the counter passes produced no rows"
        return db, ("measured in this run: two child passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of `bench.py --steps 3 --warmup 1` after the timed "
                    f"region ({time.perf_counter() - t0:.0f} s), bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md (tools/pmc_traffic.py)")
    except Exception as e:   # noqa: BLE001  (a profiler problem must not lose the headline line)
        return None, f"live PMC passes failed: {str(e)[:160]}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def flops_per_interaction(cfg, B, attention_density=1.0):
    """fwd+bwd, metadata projection once per step; attention dense (the upper bound the roofline fraction is
    quoted on) or scaled by the share of (query, key) pairs the packed users' mask allows (the "useful" figure)."""
    L, D, I, S = cfg["num_layers"], cfg["embed_dim"], cfg["intermediate_dim"], cfg["max_sequence_length"]
    V = cfg["vocab_sizes"]["0_matchedid"] + cfg["vocab_sizes"]["1_matchedid"]
    M, K = cfg["metadata_emb_size"], cfg["mask_topk"]
    return (36 * L * (D * D + D * I) + 56 * S * D * L * attention_density + 6 * K * V * D / S + 4 * V * M * D / (B * S)
            + 6 * (D * D + D) * 2 * K / S)


def extra_leg(ra, synth, cfg_name, dtype, rows, warmup, steps, table_shard=None):
    """ms per step and interactions/s of one more configuration: resident batch, no per-kernel events, one GPU
    (table_shard = (0, 1): the row-sharded item table's code path -- plan, row exchange, vocabulary-parallel heads -- at a world of one)"""
    from recommendersystem_amd.train import WSDScheduler, LambdaLR
    cfg = synth.make_config(cfg_name)
    if table_shard is not None:
        cfg["table_shard"] = table_shard
    S = cfg["max_sequence_length"]
    model = ra.RecommenderModel(cfg, device=0, dtype=dtype, max_rows=rows)
    try:
        model.init_weights(0x1217)
        model.random_pretrained_embeddings(0x3E7A)
        opt = ra.create_optimizer(model, cfg)
        sched = LambdaLR(WSDScheduler(warmup_steps=2000, total_steps=250000, decay_ratio=0.1, final_ratio=0.1))
        for _ in range(2000):
            sched.step()
        model.set_loss_weights(ra.make_task_weights(), 1)
        model.mask_seed = 0x3A5C
        model.upload(synth.make_batch(cfg, rows, 0xD47A, mu=4.6, sigma=1.0))

        def step():
            model.forward_resident(False)
            opt.step(lr_factor=sched.factor(), clip_max_norm=1.0, grad_div=1.0)
            sched.step()
        for _ in range(warmup):
            step()
        ra.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ra.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        losses = model.losses(False)
        assert all(np.isfinite(losses)), losses
        fpi = flops_per_interaction(cfg, rows)
        return {"ms_per_step": round(ms, 3), "interactions_per_sec": round(rows * S / (ms * 1e-3), 1), "steps": steps, "warmup": warmup, "dtype": dtype,
                "rows_per_gpu": rows, "workload": f"{cfg_name}: D={cfg['embed_dim']} L={cfg['num_layers']} S={S} K={cfg['mask_topk']}" + (" + row-sharded item table at world 1" if table_shard is not None else ""),
                "step_mfma_frac": round(rows * S / (ms * 1e-3) * fpi / (MFMA_PEAK_TFLOPS * 1e12), 4)}
    finally:
        model.close()


def hdf5_loop_leg(ra, synth, rows, files=6, per_file=262144):
    """The reference's input path under the driver's clock (transformer.py:37-98,633-646): train_epoch fed by
    data.PretrainDataset behind data.Prefetch from blosc-3 HDF5 shard files in the reference's layout (`file[k, blosc = 3] = v`, the format
    shards.save_data writes, through the same h5.write_h5) holding the synthetic corpus; decode + per-file block shuffle + batch cut on
    the host, one packed upload and the loss read-back per step.  cfg-3, bf16, `files` x `per_file` interactions = one epoch timed."""
    import shutil
    import tempfile
    from recommendersystem_amd import data, h5
    from recommendersystem_amd.train import ConstantScheduler, LambdaLR, train_epoch
    cfg = synth.make_config("cfg3")
    S = cfg["max_sequence_length"]
    tmp = tempfile.mkdtemp(prefix="rsys_bench_h5_")
    model = None
    try:
        base = synth.make_stream(cfg, per_file, 1)
        for sub, n in (("warm", 1), ("training", files)):
            os.makedirs(f"{tmp}/{sub}/1")
            for p in range(n):
                d = {k: v.copy() for k, v in base.items()}
                d["userid"] = np.where(d["userid"] > 0, d["userid"] + p * 100000, 0).astype(np.int32)   # other users per file, same shapes
                h5.write_h5(f"{tmp}/{sub}/1/{p}.h5", d, blosc=3)
        model = ra.RecommenderModel(cfg, device=0, dtype="bf16", max_rows=rows)
        model.init_weights(0x1217)
        model.random_pretrained_embeddings(0x3E7A)
        opt = ra.create_optimizer(model, cfg)
        sched = LambdaLR(ConstantScheduler())
        tw = ra.make_task_weights()
        train_epoch(model, data.Prefetch(data.PretrainDataset(f"{tmp}/warm", 0, 1, rows * S, seed=2)), opt, sched, tw, 1, None)
        ra.synchronize()
        # the clock starts when the producer has delivered its first batch: decoding the epoch's FIRST file (~60 ms) is a start-up cost of
        # an epoch of thousands of steps, not a per-step one (spread over this leg's 48 steps it read as +1.2 ms per step)
        import itertools
        it = iter(data.Prefetch(data.PretrainDataset(f"{tmp}/training", 0, 1, rows * S, seed=3)))
        first = next(it)
        seen = []

        def tee(src):
            for b in src:
                seen.append(b)
                yield b
        t0 = time.perf_counter()
        train_epoch(model, tee(itertools.chain([first], it)), opt, sched, tw, 1, None)
        ra.synchronize()
        n = len(seen)
        ms = (time.perf_counter() - t0) / n * 1e3
        # The same batches again, (a) from memory through the same train_epoch (what the reader thread costs) and (b) each made resident and
        # stepped without upload or read-back (what the loop costs): a step's time depends on its batch (attention tiles, selected
        # positions), so "over the resident step" is only meaningful on the same data and model.
        t1 = time.perf_counter()
        train_epoch(model, seen, opt, sched, tw, 1, None)
        ra.synchronize()
        ms_mem = (time.perf_counter() - t1) / n * 1e3
        model.set_loss_weights(tw, 1)
        res_s, res_n = 0.0, 0
        for b in seen[:: max(1, n // 8)][:8]:
            model.upload(b)
            for k in range(5):
                if k == 1:
                    ra.synchronize(); t2 = time.perf_counter()
                model.forward_resident(False)
                opt.step(lr_factor=sched.factor(), clip_max_norm=1.0)
            ra.synchronize()
            res_s += time.perf_counter() - t2; res_n += 4
        ms_res = res_s / res_n * 1e3
        return {"ms_per_step": round(ms, 3), "interactions_per_sec": round(rows * S / (ms * 1e-3), 1), "steps": n, "dtype": "bf16", "rows_per_gpu": rows,
                "same_batches_from_memory_ms_per_step": round(ms_mem, 3), "same_batches_resident_ms_per_step": round(ms_res, 3),
                "over_resident_step_pct": round((ms / ms_res - 1.0) * 100.0, 2), "over_in_memory_loop_pct": round((ms / ms_mem - 1.0) * 100.0, 2),
                "workload": f"cfg3 train_epoch from {files} blosc-3 HDF5 shard files of {per_file} interactions behind the prefetch thread; resident = 8 of the "
                            f"same batches, 4 timed steps each, same model"}
    finally:
        if model is not None:
            model.close()
        shutil.rmtree(tmp, ignore_errors=True)


def _lib_switches():
    """RSYS_* environment switches in effect that differ from the library's defaults (csrc/switches.hpp)"""
    from recommendersystem_amd import _lib
    return _lib.switches()


def git_blob_id(path):
    """the id `git hash-object` gives the file (the GPU box has no .git): which committed PMC summary a line quotes"""
    import hashlib
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def attention_tile_density(userid, S, tile=64):
    """share of the (tile x tile) blocks of a row's (2S)^2 score matrix that hold an allowed (query, key) pair under the same-user
    predicate (model.py:479-480) -- the blocks the block-sparse attention kernels visit (the token-mask predicate of :481-487
    empties almost none of them; per-wave skipping inside a visited block is not counted: an upper bound on executed work)"""
    u = np.repeat(np.asarray(userid).reshape(-1, S), 2, axis=1)
    T = 2 * S
    nt = (T + tile - 1) // tile
    vis = 0
    for row in u:
        lo = np.array([row[i * tile:(i + 1) * tile].min() for i in range(nt)])
        hi = np.array([row[i * tile:(i + 1) * tile].max() for i in range(nt)])
        # users are contiguous runs inside a row (the packer, train.py:91-98): two tiles share a user iff their id sets intersect
        sets = [set(row[i * tile:(i + 1) * tile].tolist()) for i in range(nt)]
        for i in range(nt):
            for j in range(nt):
                if lo[i] <= hi[j] and lo[j] <= hi[i] and not sets[i].isdisjoint(sets[j]):
                    vis += 1
    return vis / float(len(u) * nt * nt)


def attention_density(userid, S):
    """sum over the packed users of a row of (tokens of the user)^2 over (2S)^2, averaged over rows: the same-user predicate of
    model.py:479-487 (the token-mask predicate removes a few percent more)"""
    u = np.asarray(userid).reshape(-1, S)
    dens = []
    for row in u:
        cuts = np.flatnonzero(np.diff(row)) + 1
        runs = np.diff(np.r_[0, cuts, S])
        dens.append(float((runs.astype(np.float64) ** 2).sum()) / (S * S))
    return float(np.mean(dens))


def cpu_baseline(cfg, seed, rows=0, budget_s=40.0):
    """The package's own C++ / OpenMP restatement of the training step (oracle/cpu_step.cpp, kind "port": fp32, blocked SGEMM
    with an AVX2 / AVX-512 micro kernel on every CPU the box grants this process (affinity mask capped by the cgroup quota), per-user attention; pinned to the numpy oracle by
    tests/test_cpu_step.py) timed on this box: ONE whole step (fused item table, forward, backward incl. the metadata
    projection gradient, clip, AdamW over all parameters) measured directly, value = rows * S / that time.  rows = 0: the GPU
    step's own 64 rows when a 4096^3 SGEMM probe predicts two steps (one untimed) inside `budget_s`, else 16 (then the per-step fixed work --
    table projection both ways, optimizer -- is amortised over fewer rows than on the GPU; the split is in `sample`)."""
    from oracle import cpu_step, model_np, synth, train_np
    L = cpu_step.lib()
    L.cpu_step_set_threads(cpu_step.host_cpus())   # one thread per CPU the box grants (cgroup quota), not per core it shows
    threads = int(L.cpu_step_threads())
    S = cfg["max_sequence_length"]
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((4096, 4096), dtype=np.float32)
    gemm_rate = 0.0
    for _ in range(3):                        # best of three: the first call also starts the thread team and faults its pages in
        t0 = time.time(); cpu_step.sgemm_nt(A, A); gemm_rate = max(gemm_rate, 2 * 4096 ** 3 / (time.time() - t0))
    del A
    D, I, Lr, K = cfg["embed_dim"], cfg["intermediate_dim"], cfg["num_layers"], cfg["mask_topk"]
    V = cfg["vocab_sizes"]["0_matchedid"] + cfg["vocab_sizes"]["1_matchedid"]
    M = cfg["metadata_emb_size"]
    fixed = 4.0 * V * M * D                                                   # table projection, forward + gradient
    per_row = S * (36.0 * Lr * (D * D + D * I) + 56.0 * S * D * Lr * 0.5) + 6.0 * K * V * D   # SURVEY 8(d), attention at half density
    if rows <= 0:
        rows = 64 if 2 * (fixed + 64 * per_row) / gemm_rate * 1.3 <= budget_s else 16
    P = {}
    for name, shape in synth.param_shapes(cfg).items():    # reference init (model.py:5-12), drawn in float32
        if name.endswith(".scale"):
            w = np.ones(shape, np.float32)
        elif name.endswith(".bias") or "periodic" in name:
            w = np.zeros(shape, np.float32)
        elif "metadata_embedding" in name:
            # N(0,1)/sqrt(M) like synth.make_metadata, but 4096 distinct rows repeated: drawing 1.2 G normals would take longer
            # than the step being timed, and the arithmetic does not depend on the values
            blk = rng.standard_normal((4096, shape[1]), dtype=np.float32) / np.float32(np.sqrt(shape[1]))
            w = np.zeros(shape, np.float32)
            for r0 in range(0, shape[0] - 1, 4096):
                n = min(4096, shape[0] - 1 - r0); w[r0:r0 + n] = blk[:n]
        else:
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.006)
            if "embedding.weight" in name:
                w[-1] = 0
        P[name] = w
    d = synth.make_batch(cfg, rows, seed + 1, mu=4.6, sigma=1.0)
    wm, rm = synth.make_masks(cfg, rows, seed + 2)
    tw = train_np.make_task_weights()
    model = cpu_step.CpuStep(cfg, P, lr=1e-4)
    del P
    def one_step():
        t0 = time.time()
        dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
        losses, _ = model.forward_backward(dm, tw)
        t_fb = time.time() - t0
        norm = model.clip_adamw()
        assert all(np.isfinite(losses)) and np.isfinite(norm)
        return time.time() - t0, t_fb
    first, _ = one_step()          # untimed: first touch of ~20 GB of work buffers (the library keeps them, like a training loop)
    total, t_fb = one_step()
    cpu_step.release()
    return {"value": rows * S / total, "unit": "interactions/sec", "cores": threads, "kind": "port",
            "sample": f"oracle/cpu_step.cpp (C++/OpenMP fp32, {threads} threads, AVX{int(L.cpu_step_isa())} SGEMM {gemm_rate / 1e9:.0f} GFLOP/s at 4096^3), "
                      f"one whole train step at {rows} rows x S={S} measured directly (second of two; the first, {first:.1f}s, also pays the "
                      f"page faults of its work buffers): {total:.1f}s = fwd+bwd {t_fb:.1f}s + clip+AdamW {total - t_fb:.2f}s; no extrapolation"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # 20 warm-up + 100 timed steps, median and p10 / p90
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the two child rocprofv3 --pmc passes (roofline.traffic then comes from the committed summary)")
    ap.add_argument("--no-train-loop", action="store_true", help="skip the train_epoch (upload + loss read-back per step) measurement")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline step (oracle/cpu_step.cpp); 0 = 64 if a GEMM probe predicts <= 30 s, else 16")
    ap.add_argument("--table-shard", action="store_true",
                    help="row-sharded item table + vocabulary-parallel cross entropy (SURVEY 8(e) cfg-4; default for --config cfg4): "
                         "rank r of N holds rows [r (V+1)/N, (r+1)(V+1)/N) of the item tables and their Adam moments")
    ap.add_argument("--sampled-softmax", type=int, default=0,
                    help="row-sharded table only: classes sampled per rank and medium for the watch heads (0 = full soft-max)")
    ap.add_argument("--deterministic", action="store_true", help="bitwise reproducible steps (rsys_model_set_deterministic): what the fixed summation order costs")
    ap.add_argument("--detail", action="store_true", help="per call-site timing table on stderr")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the short legs at other configurations (cfg-2, the reference's production shape in bf16 and fp8) that the default line carries in `other_configs`")
    ap.add_argument("--zero1", action="store_true",
                    help="(needs a communicator: --gpus N or --rehearse-comm) ZeRO-1: reduce-scatter of the gradient, AdamW on this rank's 1/N of the "
                         "parameters, all-gather -- instead of the bucketed all-reduce overlapped with the backward (DESIGN 7)")
    ap.add_argument("--split-table-reduce", action="store_true",
                    help="(needs a communicator, replicated table, bf16) reduce the item table's gradient in two parts: the heads' part out of place "
                         "under the trunk backward, the batch's token rows as gathered lists in the tail (DESIGN 7)")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="1 GPU only: issue the RCCL gradient all-reduce at world size 1 (what the data-parallel step enqueues)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process has made no GPU call (it has not even loaded the library); it starts
        # N fresh ranks of itself, relays rank 0's JSON line (rank 0 inherits stdout) and exits with the first failure
        from recommendersystem_amd.dist import launch_local
        sys.exit(launch_local(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    # the contract is ONE JSON line on stdout: native libraries (RCCL prints a banner at communicator creation) write to
    # file descriptor 1 behind Python's back, so everything but that line is sent to stderr at the descriptor level
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if os.environ.get("RSYS_LIB_PATH"):        # same-box A/B of a compile-time variant of the library (tools/); the line then says so
        from recommendersystem_amd import _lib as _rlib
        _rlib.LIB_PATH = os.environ["RSYS_LIB_PATH"]
    import recommendersystem_amd as ra
    from recommendersystem_amd import workload as synth   # configurations + synthetic corpus (inputs only)
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd.train import WSDScheduler, LambdaLR, train_epoch

    rank, world, local_rank = rdist.env_rank()
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    hg = rdist.HostGroup(rank, world)
    device = local_rank if world > 1 else 0
    if os.environ.get("RSYS_BENCH_DEVICE") is not None:   # rehearsal of the N-rank path on a box with fewer GPUs
        device = int(os.environ["RSYS_BENCH_DEVICE"])
    # every rank must own a GPU: agree BEFORE any collective that could leave the others waiting
    have = 1.0 if device < ra.device_count() else 0.0
    if hg.all_reduce([have], "min")[0] < 1.0:
        print(f"bench.py rank {rank}: {ra.device_count()} GPU(s) visible, {world} ranks need one each" +
              ("" if have else f" (this rank wanted device {device})"), file=sys.stderr)
        hg.close()
        sys.exit(3)
    over = {} if args.layers is None else {"num_layers": args.layers}
    cfg = synth.make_config(args.config, **over)
    sharded = args.table_shard or args.config == "cfg4"
    if sharded:
        cfg["table_shard"] = (rank, world)
        if args.sampled_softmax:
            cfg["sampled_softmax"] = args.sampled_softmax
    S = cfg["max_sequence_length"]
    rows = args.rows
    if args.deterministic:
        cfg["deterministic"] = True
    model = ra.RecommenderModel(cfg, device=device, dtype=args.dtype, max_rows=rows)
    model.init_weights(0x1217)                 # same seed on every rank (replaces DDP's rank-0 broadcast, C1)
    model.random_pretrained_embeddings(0x3E7A)
    opt = ra.create_optimizer(model, cfg)
    comm = rdist.make_comm(hg, device)      # RCCL over xGMI (hardware_check-style self test inside)
    if comm is None and args.rehearse_comm:
        os.environ["RSYS_FORCE_RCCL"] = "1"
        comm = rdist.Comm(hg, device)
    if sharded and comm is not None:
        model.set_shard_comm(comm)             # row exchange + vocabulary-parallel head run on this communicator
    # DDP broadcasts rank 0's parameters when it wraps the model (transformer.py:678-682); here every rank initialised from the same seed
    # and the ranks compare device-side checksums of their parameter buffers: now, and again behind the timed steps
    replicas = {"after_init": rdist.assert_replicas_equal(model, comm, "after init") is not None} if (comm is not None and world > 1) else None
    sched = LambdaLR(WSDScheduler(warmup_steps=2000, total_steps=250000, decay_ratio=0.1, final_ratio=0.1))
    for _ in range(2000):
        sched.step()                           # bench at the stable learning rate
    model.set_loss_weights(ra.make_task_weights(), 1)
    model.mask_seed = 0x3A5C ^ rank
    d = synth.make_batch(cfg, rows, 0xD47A ^ rank, mu=4.6, sigma=1.0)
    model.upload(d)                            # inputs resident in HBM before the timed region
    tw = ra.make_task_weights()

    zero1 = bool(args.zero1) and comm is not None and not sharded
    if args.zero1 and not zero1:
        raise SystemExit("--zero1 needs a communicator (--gpus N or --rehearse-comm) and the replicated table")
    if zero1:
        opt.enable_zero1(comm)
    split_table = bool(args.split_table_reduce) and comm is not None and not sharded and not zero1 and args.dtype == "bf16"
    if args.split_table_reduce and not split_table:
        raise SystemExit("--split-table-reduce needs a communicator (--gpus N or --rehearse-comm), the replicated table, bf16, and no --zero1")
    if split_table:
        model.set_split_table_reduce(True)
        model.upload(d)                        # (staged again: the host counts the batch's distinct item ids only while the split reduce is on)

    def step():
        if comm is not None and not zero1:
            comm.begin_grad_sync(model)        # trunk gradient buckets are reduced while the backward runs
        model.forward_resident(False)
        if comm is not None and not zero1:
            comm.all_reduce_grads(model)
        opt.step(lr_factor=sched.factor(), clip_max_norm=1.0, grad_div=float(world))   # (zero1: reduce-scatter, partial AdamW and all-gather inside)
        sched.step()

    # The per-call-site breakdown (every kernel of a step between two HIP events on its own stream: ms_per_step_by_phase, gemm_variants,
    # hbm_kernels, executed FLOPs) is taken on the LAST warm-up steps, outside the timed region: ~240 event records cost a step 7 %.  Inside
    # the timed region only the dominant kernel family's call sites are timed (the roofline object's launch duration; ~0.4 ms per
    # instrumented step).  Fewer than three warm-up steps, or --detail: everything is timed inside the region, as before round 5.
    pre = (not args.no_kernel_timing) and (not args.detail) and args.warmup >= 3
    n_pre = min(3, args.warmup - 2) if pre else 0
    rep = {}

    def collect(into):
        # collected after every instrumented step: the event pool is reused, so its size does not depend on the shape
        # (at the production shape five steps' worth of pending events exceeded what the runtime would record)
        for k, v in model.timing_report().items():
            a = into.setdefault(k, {"ms": 0.0, "count": 0, "flops": 0.0})
            a["ms"] += v["ms"]; a["count"] += v["count"]; a["flops"] += v["flops"]

    for w in range(args.warmup):
        if pre and w == args.warmup - n_pre:
            model.timing(True, serialize=True)
        step()
        if pre and w >= args.warmup - n_pre:
            collect(rep)
    if pre:
        model.timing(False)
    ra.synchronize()
    first_losses = model.losses(False)
    hg.barrier()
    # (at least five instrumented steps once 20 are timed: two samples left the per-kernel entries swinging by 30 %; --detail: at most 20)
    n_instr = 0 if args.no_kernel_timing else (min(args.steps, 20) if args.detail else (max(5, args.steps // 10) if args.steps >= 20 else max(1, args.steps // 10)))
    dom_pre = None
    if pre and rep:
        fam = {}
        for tag, r in rep.items():
            if tag.startswith("gemm_") and "@" in tag:
                fam[tag.split("@")[1]] = fam.get(tag.split("@")[1], 0.0) + r["ms"]
        dom_pre = max(fam, key=fam.get) if fam else None
    if n_instr:
        model.timing(True, serialize=True)
        if dom_pre is not None:
            model.timing_filter("@" + dom_pre)
    ra.synchronize()
    t0 = time.perf_counter()
    rep_t = {} if pre else rep          # what the timed region's events measured (pre: the dominant family's sites only)
    n_rep = n_pre if pre else n_instr   # steps `rep` (the full breakdown) covers
    model.step_mark()
    for i in range(args.steps):
        step()
        model.step_mark()                      # an event on the compute stream per step boundary, no host sync
        if n_instr and i < n_instr:
            if pre:
                # only the dominant family's sites carry events here (~130 per step): they stay recorded and are read AFTER the timed
                # region -- reading them per step is a host wait at a step boundary (~0.3 ms of idle device each)
                if i + 1 == n_instr:
                    model.timing_pause()
            else:
                collect(rep_t)
                if i + 1 == n_instr:
                    model.timing(False)
    ra.synchronize()
    hg.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = hg.all_reduce([elapsed], "max")[0]
    if pre and n_instr:
        collect(rep_t)
        model.timing(False)
    sched_log = comm.grad_schedule(model) if (comm is not None and hasattr(comm, "grad_schedule") and not zero1) else None
    if replicas is not None:   # (outside the timed region: every rank still holds the same parameters after the timed steps' all-reduces)
        replicas["after_timed_steps"] = rdist.assert_replicas_equal(model, comm, f"after {args.warmup + args.steps} data-parallel steps") is not None
    per_step = model.step_times_ms()
    plain = per_step[n_instr:] if len(per_step) > n_instr else per_step     # steps without per-kernel events
    # the reference's real loop (train.py:238-283) beside the resident-batch number: every step uploads its batch from an
    # in-memory shard (to_device) and reads its losses back (the host sync train_epoch does), through train.train_epoch
    loop_ms = None
    if not args.no_train_loop:
        # (the resident batch itself, uploaded again every step: a step's time depends on its batch, so the loop is held against the
        # resident-batch step on the same data)
        loader = [d for i in range(max(4, args.steps))]
        train_epoch(model, loader[:2], opt, sched, tw, 1, comm)             # warm-up of this path
        ra.synchronize(); hg.barrier()
        t1 = time.perf_counter()
        train_epoch(model, loader, opt, sched, tw, 1, comm)
        ra.synchronize(); hg.barrier()
        loop_ms = hg.all_reduce([(time.perf_counter() - t1) / len(loader) * 1e3], "max")[0]
    # exposed (non-overlapped) all-reduce time: the same resident-batch step with the gradient all-reduce left
    # out.  Measured LAST: from here on the ranks' parameters drift apart, and nothing that is reported runs afterwards.
    nocomm_ms = None
    if comm is not None and not sharded:
        model.upload(d)

        def step_nc():
            model.forward_resident(False)
            opt.step(lr_factor=sched.factor(), clip_max_norm=1.0, grad_div=1.0)
        for _ in range(3):
            step_nc()
        ra.synchronize(); hg.barrier()
        t2 = time.perf_counter()
        k_nc = max(10, args.steps // 5)
        for _ in range(k_nc):
            step_nc()
        ra.synchronize(); hg.barrier()
        nocomm_ms = hg.all_reduce([(time.perf_counter() - t2) / k_nc * 1e3], "max")[0]
    if rep:
        # the head GEMMs stop at the positive-weight rows (device-side limit): count the flops they really did
        npos = model.head_rows()
        D = cfg["embed_dim"]; V0 = cfg["vocab_sizes"]["0_matchedid"]; V1 = cfg["vocab_sizes"]["1_matchedid"]
        up = lambda n, q: (n + q - 1) // q * q
        fl_rows = sum(2.0 * up(npos[2 * m_], 128) * v * D for m_, v in ((0, V0), (1, V1))) * n_rep
        fl_k = sum(2.0 * up(npos[2 * m_], 64) * v * D for m_, v in ((0, V0), (1, V1))) * n_rep
        fl_rows256 = sum(2.0 * up(npos[2 * m_], 256) * v * D for m_, v in ((0, V0), (1, V1))) * n_rep
        for tag, fl in (("gemm_logits", fl_rows), ("gemm_head_dx", fl_rows), ("gemm_head_dw", fl_k)):
            for full in [k for k in rep if k.split("@")[0] == tag]:
                rep[full]["flops"] = fl_rows256 if (tag != "gemm_head_dw" and full.endswith(("@8p", "@8c"))) else fl   # 256-row tiles
        # the rating heads stop at their live rows too (tasks 1 and 3; one tag covers both launches)
        KBr = cfg["mask_topk"] * rows
        for tag, q in (("gemm_rating_fwd", 128), ("gemm_rating_dx", 128), ("gemm_rating_dw", 64)):
            for full in [k for k in rep if k.split("@")[0] == tag]:
                rep[full]["flops"] *= (min(KBr, up(npos[1], q)) + min(KBr, up(npos[3], q))) / (2.0 * KBr)
        # compact top of the trunk: those GEMMs stop at the selected tokens (device-side limit), their tags carry the capacity
        try:
            top_cap = int(model.debug_get("top.cap", rows)[0])
            top_n = int(model.debug_get("top.n", rows)[0]) if top_cap else 0
        except Exception:
            top_cap, top_n = 0, 0
        if top_cap:
            for full in [k for k in rep if k.startswith("gemm_top_")]:
                q = 256 if full.endswith(("@8p", "@8c")) else (64 if full.split("@")[0].endswith("_dw") else 128)
                rep[full]["flops"] *= min(top_cap, up(top_n, q)) / float(top_cap)
    losses = model.losses(False)
    assert all(np.isfinite(losses)), losses
    # which collectives library ran and what the step's gradient reduction enqueued (a SCALE record then explains itself: DESIGN 7 holds
    # the N = 8 prediction these entries are to be read against)
    comm_info = None
    if comm is not None and hasattr(comm, "info"):
        comm_info = dict(comm.info(), kind=type(comm).__name__)
        comm_info.pop("rank", None)
        if sched_log is not None:
            names = {0: "early bucket (inside the backward)", 1: "tail beside the dWp GEMM", 2: "dWp", 3: "split table reduce: head part, out of place",
                     4: "split table reduce: gathered token rows"}
            comm_info["bucket_schedule"] = [{"phase": names.get(ph, str(ph)), "MB": round((hi - lo) * 4 / 1e6, 1)} for lo, hi, ph in sched_log]
    elif comm is not None:
        comm_info = {"kind": type(comm).__name__, "ranks": world}

    if rank == 0:
        inter = world * rows * S * args.steps
        value = inter / elapsed
        ms = elapsed / args.steps * 1e3
        fpi = flops_per_interaction(cfg, rows)
        dens = attention_density(d["userid"], S)
        fpi_useful = flops_per_interaction(cfg, rows, dens)
        # FLOPs the kernels really execute per step: every GEMM call site at its device-side limits (head GEMMs at the live rows,
        # the compact top at the selected tokens) + the attention kernels at the 64 x 64 blocks they visit (2 products forward,
        # 7 backward: the dK/dV and the dQ kernel each recompute S and dP)
        executed = None
        if rep:
            hd = cfg["embed_dim"] // cfg["num_heads"]
            tdens = attention_tile_density(d["userid"], S)
            attn_fl = cfg["num_layers"] * cfg["num_heads"] * rows * (2 * S) ** 2 * tdens * (2 + 7) * 2.0 * hd
            gemm_fl = sum(r["flops"] for tag, r in rep.items() if tag.startswith("gemm_")) / n_rep
            executed = {"gemm_flops_per_step": gemm_fl, "attention_flops_per_step": attn_fl, "attention_tile_density": round(tdens, 4)}
        # dominant kernel: the MFMA GEMM family, per instantiation
        var = {}
        for tag, r in rep.items():
            v = tag.split("@")[1] if tag.startswith("gemm_") and "@" in tag else None
            if v:
                a = var.setdefault(v, {"ms": 0.0, "flops": 0.0, "launches": 0})
                a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["launches"] += r["count"]
        roofline = None
        traffic_db = {}
        # PMC passes are separate runs (rocprofv3 --pmc) of the default workload; their per-launch summary is committed under
        # profiles/ and only describes that workload
        pmc_applies = args.config == "cfg3" and rows == 64 and args.layers is None and args.dtype == "bf16" and not sharded and not args.deterministic
        traffic_note = None
        live_ok = (pmc_applies and world == 1 and rows == 64 and not args.no_live_pmc and not args.detail and not args.no_kernel_timing
                   and not zero1 and not split_table and comm is None)
        if live_ok:
            traffic_db, traffic_note = live_pmc_traffic()
            traffic_db = traffic_db or {}
        traffic_live = bool(traffic_db)
        try:
            if pmc_applies and not traffic_db:
                traffic_db = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))["kernels"]
        except Exception:
            pass
        if var:
            dom = max(var, key=lambda k: var[k]["ms"])
            a = dict(var[dom])
            if pre:
                # duration from the TIMED region's events (the family's call sites only); FLOPs per step from the full breakdown of the
                # warm-up steps (the same resident batch every step: the same launches and device-side limits)
                t_ms = sum(r["ms"] for tag, r in rep_t.items() if tag.endswith("@" + dom))
                t_n = sum(r["count"] for tag, r in rep_t.items() if tag.endswith("@" + dom))
                if t_ms > 0 and t_n > 0 and n_instr:
                    a = {"ms": t_ms, "launches": t_n, "flops": a["flops"] / n_rep * n_instr}
            ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS_FP8 if dom in ("8f", "8fs", "8gf") else MFMA_PEAK_TFLOPS
            roofline = {"bound": "mfma", "kernel": KERNEL_LABEL.get(dom, dom), "achieved": round(ach, 1),
                        "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": traffic_of(traffic_db, dom),
                        "traffic_source": (None if not traffic_db else {"note": traffic_note} if traffic_live else
                                           {"file": "profiles/" + TRAFFIC_FILE, "git_blob": git_blob_id(os.path.join(ROOT, "profiles", TRAFFIC_FILE)), "live_attempt": traffic_note,
                                                                        "note": "separate rocprofv3 --pmc passes of this workload (tools/prof_round.sh), read from the committed summary, not measured in this run"}),
                        "avg_launch_ms": round(a["ms"] / a["launches"], 4),
                        "launches": a["launches"],
                        "share_of_step": round(a["ms"] / max(n_instr, 1) / ms, 3), "instrumented_steps": n_instr,
                        "timed_with": ("HIP events around this family's call sites only, first %d of the %d timed steps; the other per-kernel fields of this line come from %d fully "
                                       "instrumented warm-up steps" % (n_instr, args.steps, n_rep)) if pre else "HIP events around every call site, first %d of the timed steps" % n_instr}
        out = {
            "metric": "interactions/sec", "value": round(value, 1), "unit": "interactions/sec",
            "user_seqs_per_sec": round(value / S, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "comm": comm_info,
            "replicas_consistent": replicas,   # per-rank parameter checksums equal on every rank (None: one rank, nothing to compare)
            "switches": _lib_switches(),     # RSYS_* switches that differ from the library's defaults ({} = the shipped path)
            "library": os.environ.get("RSYS_LIB_PATH") or "recommendersystem_amd/librsys_hip.so",
            "config": {"workload": f"{args.config}: train step fwd+bwd+{'allreduce+' if comm is not None else ''}clip+AdamW, D={cfg['embed_dim']} L={cfg['num_layers']} "
                                   f"S={S} V={cfg['vocab_sizes']['0_matchedid'] + cfg['vocab_sizes']['1_matchedid']} M={cfg['metadata_emb_size']} "
                                   f"K={cfg['mask_topk']}", "rows_per_gpu": rows, "global_rows": rows * world,
                       "parallelism": f"dp{world}" + (f" + item table row-sharded x{world} (vocab-parallel CE, sparse row exchange" +
                                                                  (f", sampled soft-max {args.sampled_softmax}/rank/medium" if args.sampled_softmax else "") + ")" if sharded else "")
                                      + (" + ZeRO-1 optimizer" if zero1 else "") + (" + split table-gradient reduce" if split_table else "")},
            "model_flops_per_interaction": fpi,
            "step_mfma_frac": round(value / world * fpi / (MFMA_PEAK_TFLOPS * 1e12), 4),
            # the same fraction on "useful" FLOPs, attention scaled by the share of same-user (query, key) pairs of this batch
            "attention_density": round(dens, 4), "model_flops_per_interaction_useful": round(fpi_useful, 1),
            "step_mfma_frac_useful": round(value / world * fpi_useful / (MFMA_PEAK_TFLOPS * 1e12), 4),
            # ... and on the FLOPs that were EXECUTED (zero-weight head rows, unselected tokens of the last layer's tail and masked
            # attention blocks are skipped, not computed): the utilisation figure; the two above price the model's nominal work
            "step_mfma_frac_executed": (None if executed is None else round((executed["gemm_flops_per_step"] + executed["attention_flops_per_step"]) / (ms * 1e-3) / (MFMA_PEAK_TFLOPS * 1e12), 4)),
            "executed": executed,
            "roofline": roofline,
            "losses": [round(float(x), 4) for x in losses],
        }
        if args.dtype in ("fp8", "float8"):
            out["dtype_note"] = ("bf16 arithmetic with the transformer blocks' linears (forward, dx, dW) on tensor-wise dynamically scaled e4m3 / e5m2 "
                                 "operands: the reference's pretraining arithmetic (transformer.py:671-676, torchao 'tensorwise'), restated -- parity "
                                 "pinned to torch's float8 casts and torch._scaled_mm, not to torchao itself (absent); step_mfma_frac* are still priced against the bf16 peak")
        if len(plain):
            q = lambda f: round(float(np.quantile(plain, f)), 3)
            out["ms_per_step_stats"] = {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "n": int(len(plain)),
                                        "note": "HIP-event time between step boundaries on the compute stream, steps without per-kernel events"}
        if nocomm_ms is not None:
            out["ms_per_step_without_allreduce"] = round(nocomm_ms, 3)
            # what the gradient all-reduce adds to a step after its overlap with the backward: both arms without per-kernel events
            # (the median of the un-instrumented timed steps against the un-instrumented no-all-reduce loop)
            with_ar = float(np.median(plain)) if len(plain) else ms
            out["allreduce_exposed_ms_per_step"] = round(with_ar - nocomm_ms, 3)
        if loop_ms is not None:
            out["train_loop_ms_per_step"] = round(loop_ms, 3)     # train_epoch: batch upload + loss read-back every step
            out["train_loop_interactions_per_sec"] = round(world * rows * S / (loop_ms * 1e-3), 1)
        hbm = {}
        for tag, r in rep.items():                                # HBM-bound row kernels: algorithmic bytes / HIP-event time
            if (tag.startswith("hbm_") or tag in ("adamw", "sumsq")) and r["ms"] > 0:
                e = {"GBps": round(r["flops"] / (r["ms"] * 1e-3) / 1e9, 1), "ms_per_step": round(r["ms"] / n_rep, 4),
                     "launches_per_step": r["count"] // n_rep, "algorithmic_MB_per_launch": round(r["flops"] / r["count"] / 1e6, 2)}
                e["frac_of_hbm_peak"] = round(e["GBps"] / HBM_PEAK_GBS, 3)
                pm = traffic_of_name(traffic_db, HBM_KERNEL_SYMBOL.get(tag))
                if pm:
                    e["pmc_MB_per_launch"] = round(pm / 1e6, 2)
                hbm[tag[4:] if tag.startswith("hbm_") else tag] = e
        if hbm:
            out["hbm_kernels"] = hbm
        if rep:
            out["per_kernel_fields_measured_on"] = (f"the last {n_rep} warm-up steps, every call site between HIP events (outside the timed region)" if pre
                                                    else f"the first {n_rep} timed steps, every call site between HIP events")
            phases = {k: round(v["ms"] / n_rep, 3) for k, v in rep.items() if k.startswith("phase_") or k in ("adamw", "sumsq", "attn_fwd", "attn_bwd", "ce")}
            out["ms_per_step_by_phase"] = phases
            out["gemm_variants"] = {k: {"ms_per_step": round(v["ms"] / n_rep, 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in var.items()}
        if args.detail:
            for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
                print(f"  {k:22s} {v['ms'] / n_rep:8.3f} ms/step  {v['count'] // n_rep:4d} launches/step  {tf:7.1f} TFLOP/s", file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, 1, args.cpu_rows)
        # the numbers README / DESIGN quote for other configurations, under the same clock as `value`: short
        # resident-batch legs after the cfg-3 measurement; `value`, `config` and everything above describe cfg-3 in bf16 only
        default_run = (world == 1 and args.config == "cfg3" and args.dtype == "bf16" and rows == 64 and args.layers is None and not sharded
                       and not args.deterministic and not args.detail and not args.no_kernel_timing)
        if default_run and not args.no_extra_legs:
            model.close(); model = None
            legs = {}
            for key, cname, dt, warm, k, shard in (("cfg2_bf16", "cfg2", "bf16", 5, 20, None), ("cfg3_fp8", "cfg3", "fp8", 5, 20, None),
                                                   ("cfg4_world1", "cfg4", "bf16", 3, 10, (0, 1)),
                                                   ("prod_bf16", "prod", "bf16", 2, 5, None), ("prod_fp8", "prod", "fp8", 3, 10, None)):
                try:
                    legs[key] = extra_leg(ra, synth, cname, dt, 64, warm, k, table_shard=shard)
                except Exception as e:   # noqa: BLE001  (a failing leg must not lose the headline line)
                    legs[key] = {"error": str(e)[:200]}
            try:
                legs["hdf5_loop_cfg3"] = hdf5_loop_leg(ra, synth, 64)
            except Exception as e:   # noqa: BLE001
                legs["hdf5_loop_cfg3"] = {"error": str(e)[:200]}
            legs["note"] = ("train step (fwd + bwd + fused clip/AdamW) on one resident synthetic batch of 64 rows per configuration, un-instrumented; "
                            "prod = the reference's production shape (D=2048 L=8 S=1024 I=5632 K=128, transformer.py:535-560, 200 K items); "
                            "fp8 = the opt-in torchao-style tensorwise trunk (parity unpinned for torchao's scale formula; the compact top is off in that mode); "
                            "cfg4_world1 = BASELINE configs[3]'s model (D=1024, row-sharded 200K x 1024 table) with every shard on this GPU; "
                            "hdf5_loop_cfg3 = the reference's loader path (blosc-3 HDF5 shards, block shuffle, prefetch thread, double-buffered upload, losses parked on "
                            "the device and read once per 1024 steps as the reference reads its device tensors once per epoch) against the SAME batches replayed from "
                            "memory and made resident on the same model (a step's time depends on its batch)")
            out["other_configs"] = legs
        print(json.dumps(out), file=json_out, flush=True)
    if comm is not None:
        comm.close()
    if model is not None:
        model.close()
    hg.close()


if __name__ == "__main__":
    main()
