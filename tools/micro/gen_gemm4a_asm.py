"""Generator of the hand-placed K loop of tools/micro/gemm4a.hip: writes tools/micro/gemm4a_asm.inc, ONE inline-asm
statement (prologue requests, the K loop unrolled by two K tiles, the bf16 store) with every register named, because hipcc cannot keep
256 accumulators in place (tools/micro/gemm4a.hip, the HIP-level kernel: 512 v_accvgpr moves per two K tiles).

Registers: a[0:255] accumulators (block (i, j) of the wave's 128 x 128 at a[(8 i + j) 4 ..+3]); v[128:255] four fragment sets A_x, A_y,
B_x, B_y (fragment (b, kk) at set + (2 b + kk) 4); s[60:87] descriptors, scalar offsets and counters; everything else through operands.
Schedule per phase (32 MFMAs): s_waitcnt lgkmcnt(0) + vmcnt(24), s_barrier, then the MFMAs with the 8 fragment reads of the NEXT phase
behind the first eight of them and the 4 LDS-DMA requests (region read a phase ago, for the K tile two ahead) behind four later ones.
    python tools/micro/gen_gemm4a_asm.py      (writes gemm4a_asm.inc and gemm4a_clobbers.inc beside itself)
"""
SET = {"Ax": 128, "Ay": 160, "Bx": 192, "By": 224}
out = []
def emit(s):
    out.append(s)

def frag(setname, b, kk):
    r = SET[setname] + ((4 * b + kk) if MFMA32 else (2 * b + kk)) * 4
    return f"v[{r}:{r + 3}]"

def acc(i, j):
    if MFMA32:                      # 4 x 4 blocks of 32 x 32, 16 registers each
        r = (4 * i + j) * 16
        return f"a[{r}:{r + 15}]"
    r = (8 * i + j) * 4
    return f"a[{r}:{r + 3}]"

def dma(kind, stage, h, j):
    """one LDS-DMA instruction of half-tile (kind, h) into `stage`: descriptor s[60:63] (A) / s[64:67] (B), scalar offset s[68 + 4 h + j] / s[76 + ..]"""
    region = (0 if kind == "A" else 2) + h
    imm = stage * 65536 + region * 16384 + j * 1024
    desc = "s[60:63]" if kind == "A" else "s[64:67]"
    soff = (68 if kind == "A" else 76) + 4 * h + j
    voff = f"%[v{kind.lower()}{j & 1}]"
    return [f"s_add_u32 m0, %[dmalds], {imm}", "s_nop 0", f"buffer_load_dwordx4 {voff}, {desc}, s{soff} offen lds"]

def reads(kind, setname, stage, h):
    res = []
    if MFMA32:
        # fragment (b, s): 32-row block b of the half, k step s (16 k): address register per k step (the swizzle XORs the chunk index),
        # stage 1 = the same + 65536 in v[112:119]
        for b in range(2):
            for ks in range(4):
                r = (0 if kind == "A" else 4) + ks
                addr = f"%[r{kind.lower()}{ks}]" if stage == 0 else f"v{112 + r}"
                res.append(f"ds_read_b128 {frag(setname, b, ks)}, {addr} offset:{h * 16384 + b * 4096}")
        return res
    for b in range(4):
        for kk in range(2):
            # stage 0: operand registers; stage 1: the same + 65536 in v[120:123] (computed below: the ds offset field has 16 bits)
            addr = f"%[r{kind.lower()}{kk}]" if stage == 0 else f"v{120 + (0 if kind == 'A' else 2) + kk}"
            res.append(f"ds_read_b128 {frag(setname, b, kk)}, {addr} offset:{h * 16384 + b * 2048}")
    return res

def advance(kind):
    lo, rem, recw, full = ("%[alo]", "s84", "s62", "%[reca]") if kind == "A" else ("%[blo]", "s85", "s66", "%[recb]")
    hi = "%[ahi]" if kind == "A" else "%[bhi]"
    d0, d1 = ("s60", "s61") if kind == "A" else ("s64", "s65")
    return [f"s_add_u32 {lo}, {lo}, 128", f"s_addc_u32 {hi}, {hi}, 0", f"s_mov_b32 {d0}, {lo}", f"s_and_b32 {d1}, {hi}, 0xffff",
            f"s_sub_i32 {rem}, {rem}, 1", f"s_cmp_gt_i32 {rem}, 0", f"s_cselect_b32 {recw}, {full}, 0"]

import sys as _sys
BARRIER_EVERY = 2     # phases per s_barrier (1: every phase, 24 requests in flight at the wait; 2: before P1 and P3, 16 in flight)
NO_DMA = "--no-dma" in _sys.argv            # timing-only variants (wrong results): what the loop costs without its requests / barriers
NO_BARRIER = "--no-barrier" in _sys.argv
MFMA32 = "--mfma32" in _sys.argv            # v_mfma_f32_32x32x16_bf16: 16 instructions of 32 cycles per phase, wider gaps for the requests
SUFFIX = "".join(a.replace("--", "_").replace("-", "") for a in _sys.argv[1:])

def phase(ih, jh, aset, bset, pref, dmas, post=(), barrier=True):
    """pref = (kind, set, stage, h) fragment reads for the next phase; dmas = (kind, stage, h) region to request; post: scalar bookkeeping"""
    emit("s_waitcnt lgkmcnt(0)")
    if barrier and not NO_BARRIER:
        if not NO_DMA:
            emit(f"s_waitcnt vmcnt({24 if BARRIER_EVERY == 1 else 16})")
        emit("s_barrier")
    rd = reads(*pref)
    dm = [dma(dmas[0], dmas[1], dmas[2], j) for j in range(4)]
    n = 0
    if MFMA32:
        dma_at = (5, 8, 11, 14)
        for i in range(2):
            for j in range(2):
                for ks in range(4):
                    n += 1
                    if n in dma_at and not NO_DMA:
                        emit(dm[dma_at.index(n)][0])
                    emit(f"v_mfma_f32_32x32x16_bf16 {acc(2 * ih + i, 2 * jh + j)}, {frag(bset, j, ks)}, {frag(aset, i, ks)}, {acc(2 * ih + i, 2 * jh + j)}")
                    if n <= 4:
                        emit(rd[2 * n - 2]); emit(rd[2 * n - 1])
                    elif n in dma_at and not NO_DMA:
                        emit(dm[dma_at.index(n)][2])
    else:
        for i in range(4):
            for j in range(4):
                for kk in range(2):
                    n += 1
                    if n in (11, 15, 19, 23) and not NO_DMA:
                        emit(dm[(n - 11) // 4][0])          # s_add_u32 m0: the MFMA below is its wait state in front of the DMA
                    emit(f"v_mfma_f32_16x16x32_bf16 {acc(4 * ih + i, 4 * jh + j)}, {frag(bset, j, kk)}, {frag(aset, i, kk)}, {acc(4 * ih + i, 4 * jh + j)}")
                    if n <= 8:
                        emit(rd[n - 1])
                    elif n in (11, 15, 19, 23) and not NO_DMA:
                        emit(dm[(n - 11) // 4][2])
    for line in post:
        emit(line)

def ktile(stage, bc, bn):
    st, so = stage, stage ^ 1
    if BARRIER_EVERY == 1:
        phase(0, 0, "Ax", bc, ("B", bn, st, 1), ("B", st, 0))
        phase(0, 1, "Ax", bn, ("A", "Ay", st, 1), ("B", st, 1), advance("B"))
        phase(1, 1, "Ay", bn, ("A", "Ax", so, 0), ("A", st, 1), advance("A"))
        phase(1, 0, "Ay", bc, ("B", bn, so, 0), ("A", so, 0))
    else:
        # requests (K tile t + 2, same stage): P1 A h0, P2 B h0 (both read in P3 / P4 of the previous K tile), P3 B h1, P4 A h1 (read in P1 / P2)
        phase(0, 0, "Ax", bc, ("B", bn, st, 1), ("A", st, 0))
        phase(0, 1, "Ax", bn, ("A", "Ay", st, 1), ("B", st, 0), barrier=False)
        phase(1, 1, "Ay", bn, ("A", "Ax", so, 0), ("B", st, 1))
        phase(1, 0, "Ay", bc, ("B", bn, so, 0), ("A", st, 1), advance("A") + advance("B"), barrier=False)

# ---------------------------------------------------------------- setup: descriptors, scalar offsets
emit("s_mov_b32 s60, %[alo]"); emit("s_and_b32 s61, %[ahi], 0xffff"); emit("s_mov_b32 s62, %[reca]"); emit("s_mov_b32 s63, 0x00020000")
emit("s_mov_b32 s64, %[blo]"); emit("s_and_b32 s65, %[bhi], 0xffff"); emit("s_mov_b32 s66, %[recb]"); emit("s_mov_b32 s67, 0x00020000")
for h in range(2):
    for j in range(4):
        emit(f"s_mul_i32 s{68 + 4 * h + j}, %[unita], {8 * h + j}")     # (h * 64 + j * 8) rows * lda * 2 bytes
        emit(f"s_mul_i32 s{76 + 4 * h + j}, %[unitb], {8 * h + j}")
if MFMA32:
    for ks in range(4):
        emit(f"v_add_u32 v{112 + ks}, 0x10000, %[ra{ks}]"); emit(f"v_add_u32 v{116 + ks}, 0x10000, %[rb{ks}]")
else:
    emit("v_add_u32 v120, 0x10000, %[ra0]"); emit("v_add_u32 v121, 0x10000, %[ra1]"); emit("v_add_u32 v122, 0x10000, %[rb0]"); emit("v_add_u32 v123, 0x10000, %[rb1]")
# accumulators start from zero
for r in range(256):
    emit(f"v_accvgpr_write_b32 a{r}, 0")
# ---------------------------------------------------------------- prologue: K tiles 0 and 1 in consumption order, then A h0 of K tile 2
emit("s_nop 4")
def request_group(kind, stage, h):
    for j in range(4):
        for line in dma(kind, stage, h, j):
            emit(line)
for stage in range(2):
    request_group("A", stage, 0); request_group("B", stage, 0); request_group("B", stage, 1); request_group("A", stage, 1)
    # next K tile: both descriptors one K tile on (the launcher guarantees nt >= 2)
    for line in advance("A")[:4] + advance("B")[:4]:
        emit(line)
# remaining valid K tiles from K tile 2 on
emit("s_sub_i32 s84, %[nt], 2"); emit("s_cmp_gt_i32 s84, 0"); emit("s_cselect_b32 s62, %[reca], 0")
emit("s_sub_i32 s85, %[nt], 2"); emit("s_cmp_gt_i32 s85, 0"); emit("s_cselect_b32 s66, %[recb], 0")
emit("s_waitcnt vmcnt(24)")
emit("s_barrier")
for line in reads("A", "Ax", 0, 0) + reads("B", "Bx", 0, 0):
    emit(line)
if BARRIER_EVERY == 1:
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    request_group("A", 0, 0)                     # A h0 of K tile 2 (what the fourth phase of "K tile -1" would have requested; A advances after P3)
emit("s_lshr_b32 s86, %[nt], 1")            # loop trips: two K tiles each
emit("1:")
ktile(0, "Bx", "By")
ktile(1, "By", "Bx")
emit("s_sub_u32 s86, s86, 1")
emit("s_cmp_lg_u32 s86, 0")
emit("s_cbranch_scc1 1b")
# ---------------------------------------------------------------- bf16 store of the 8 x 8 blocks: row offsets in s[68:75] (re-used), columns as immediates
emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
emit("s_nop 15"); emit("s_nop 15")
if MFMA32:
    # block (bi, bj): lane (m = l & 31, hh = l >> 5) holds C[32 bi + m][32 bj + 8 g + 4 hh + 0..3] in registers 4 g .. 4 g + 3; %[vc] = the lane's
    # (row m, column 4 hh) offset, s[68 + bi] = 32 bi rows, the immediate = (32 bj + 8 g) columns
    for i in range(4):
        emit(f"s_mul_i32 s{68 + i}, %[unitc], {2 * i}")
    cnt = 0
    for i in range(4):
        for j in range(4):
            for g in range(4):
                t = 128 + (cnt % 8) * 8; cnt += 1
                r = (4 * i + j) * 16 + 4 * g
                for k in range(4):
                    emit(f"v_accvgpr_read_b32 v{t + k}, a{r + k}")
                emit(f"v_cvt_pk_bf16_f32 v{t + 4}, v{t}, v{t + 1}")
                emit(f"v_cvt_pk_bf16_f32 v{t + 5}, v{t + 2}, v{t + 3}")
                emit(f"buffer_store_dwordx2 v[{t + 4}:{t + 5}], %[vc], %[cdesc], s{68 + i} offen offset:{(32 * j + 8 * g) * 2}")
else:
  for i in range(8):
    emit(f"s_mul_i32 s{68 + i}, %[unitc], {i}")      # 16 i rows * ldc * 2 bytes
  for i in range(8):
    for j in range(8):
        t = 128 + ((i * 8 + j) % 8) * 8
        r = (8 * i + j) * 4
        for k in range(4):
            emit(f"v_accvgpr_read_b32 v{t + k}, a{r + k}")
        emit(f"v_cvt_pk_bf16_f32 v{t + 4}, v{t}, v{t + 1}")
        emit(f"v_cvt_pk_bf16_f32 v{t + 5}, v{t + 2}, v{t + 3}")
        emit(f"buffer_store_dwordx2 v[{t + 4}:{t + 5}], %[vc], %[cdesc], s{68 + i} offen offset:{j * 32}")
emit("s_waitcnt vmcnt(0)")

import os, sys
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, f"gemm4a_asm{SUFFIX}.inc"), "w") as f:
    f.write("// generated by tools/micro/gen_gemm4a_asm.py -- do not edit\n")
    for l in out:
        f.write(f'"{l}\\n\\t"\n')
with open(os.path.join(here, "gemm4a_clobbers.inc"), "w") as f:
    f.write("// generated by tools/micro/gen_gemm4a_asm.py -- do not edit\n")
    regs = [f"v{r}" for r in range(112, 256)] + [f"a{r}" for r in range(256)] + [f"s{r}" for r in range(60, 88)]
    f.write(", ".join(f'"{r}"' for r in regs) + "\n")
print(len(out), "asm lines")
