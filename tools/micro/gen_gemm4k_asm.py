"""Generator of the K-MAJOR member of the four-wave register-named GEMM loops (gemm4p's sibling): C[M][N] += sum_k A[k][m] B[k][n], bf16 operands stored
[K][ld] (the weight gradients dW = dY^T X, K = tokens), one (output tile, K split) per workgroup, fp32 atomics.  Writes gemm4k_asm.inc = ONE asm
statement: prologue requests, the K loop unrolled by two K tiles, the atomic epilogue.

LDS image, request pattern and fragment reads are gemm8p.hip's K-major form on gemm4p's wave layout (2 x 2 waves of 128 x 128):
  * a 16 KB half-tile region (A h: the 64 columns [64 h, 64 h + 64) of both wave rows) = [64 k][256 B]; the 16-column block mb of row k sits in block
    slot mb ^ f(k), f(k) = ((k >> 3) & 1) << 2 | (k & 3).  Wave w requests the 1 KB pieces 4 w .. 4 w + 3 (k rows 16 w + 4 j + (l >> 4), 16-byte chunk
    l & 15): per-lane source offset by j >> 1 (the swizzle), the 4 j rows and the half as scalar offsets;
  * a fragment (16 columns x 8 k of a 32-wide step) = two ds_read_b64_tr_b16, 1 KB apart (k rows + 4): registers r, r + 1 and r + 2, r + 3;
  * windows (s[88:91], num_records s62 / s66 = the BYTES left in this workgroup's K range): K rows past the range read zeros, so an odd K-tile
    count and a K that is no multiple of 64 need no code -- the loop always runs pairs.
Schedule per phase (32 MFMAs): lgkmcnt(0) [+ vmcnt(16), barrier in P1 / P3], the 16 fragment reads of the NEXT phase behind the first 16 MFMAs, the 4
requests (region read a phase ago, K tile two ahead) behind four later ones.
Epilogue: each wave passes its 128 x 128 block through a private LDS patch, 32 rows at a time ([32][132] f32), so that every atomic wave instruction
adds 256 contiguous bytes (the full-rate shape); rows past M fall outside the output descriptor, columns past N are EXEC-masked.
Registers: a[0:255] accumulators; v[128:255] fragment sets A_x, A_y, B_x, B_y (epilogue: temporaries); v[104:119] fragment read addresses
(operand x block x stage); v[96:103] request offsets (rotating); s[60:67] descriptors, s[68:83] scalar row offsets, s[84:87] byte counters / trip count, s[88:91] windows, s[92:95] epilogue.
    python tools/micro/gen_gemm4k_asm.py [outdir]   (default: recommendersystem_amd/csrc; the committed copies are compared by tests/test_host_logic.py)
"""
import os
import sys

NO_MFMA = "--no-mfma" in sys.argv     # timing-only variant
SET = {"Ax": 128, "Ay": 160, "Bx": 192, "By": 224}
DMA_AT = (18, 22, 26, 30)
out = []


def emit(s):
    out.append(s)


def frag(setname, b, kk):
    return SET[setname] + (2 * b + kk) * 4


def acc(i, j):
    r = (8 * i + j) * 4
    return f"a[{r}:{r + 3}]"


_tmp = [0]


def dma(kind, stage, h, j):
    """one LDS-DMA request.  The 4 j rows and the half's columns are added to the per-lane offset (a rotating scratch register) rather than passed as the
    instruction's scalar offset: the windows here end INSIDE a K tile (a K that is no multiple of 64), and this way the cut does not depend on how the range
    check treats the scalar offset.  (Both forms pass test_gemm_kmajor_splitk_stops_at_k on gfx950 and run at the same speed: the check does cover it.)"""
    region = (0 if kind == "A" else 2) + h
    imm = stage * 65536 + region * 16384 + j * 1024
    desc = "s[60:63]" if kind == "A" else "s[64:67]"
    soff = (68 if kind == "A" else 76) + 4 * h + j
    voff = f"%[v{kind.lower()}{j >> 1}]"
    t = 96 + _tmp[0]; _tmp[0] = (_tmp[0] + 1) % 8
    return [f"s_add_u32 m0, %[dmalds], {imm}", f"v_add_u32 v{t}, s{soff}, {voff}", f"buffer_load_dwordx4 v{t}, {desc}, 0 offen lds"]


def reads(kind, setname, stage, h):
    """the 16 transposed reads of the four fragments pairs (block b, k step kk) of half-tile (kind, h) of `stage`"""
    res = []
    base = (104 if kind == "A" else 112) + 4 * stage
    for b in range(4):
        for kk in range(2):
            r = frag(setname, b, kk)
            off = (0 if kind == "A" else 32768) + h * 16384 + kk * 8192
            res.append(f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{base + b} offset:{off}")
            res.append(f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], v{base + b} offset:{off + 1024}")
    return res


def advance(kind):
    """window of `kind` one K tile on; num_records = the bytes left (0 behind the range)"""
    lo, hi, rem, recw, step = ("s88", "s89", "s84", "s62", "%[stepa]") if kind == "A" else ("s90", "s91", "s85", "s66", "%[stepb]")
    d0, d1 = ("s60", "s61") if kind == "A" else ("s64", "s65")
    return [f"s_add_u32 {lo}, {lo}, {step}", f"s_addc_u32 {hi}, {hi}, 0", f"s_mov_b32 {d0}, {lo}", f"s_and_b32 {d1}, {hi}, 0xffff",
            f"s_sub_u32 {rem}, {rem}, {step}", f"s_cselect_b32 {rem}, 0, {rem}", f"s_mov_b32 {recw}, {rem}"]


def phase(ih, jh, aset, bset, pref, dmas, post=(), barrier=True):
    emit("s_waitcnt lgkmcnt(0)")
    if barrier:
        emit("s_waitcnt vmcnt(16)")
        emit("s_barrier")
    rd = reads(*pref)
    dm = [dma(dmas[0], dmas[1], dmas[2], j) for j in range(4)]
    n = 0
    for i in range(4):
        for j in range(4):
            for kk in range(2):
                n += 1
                if n in DMA_AT:
                    emit(dm[DMA_AT.index(n)][0])        # m0 and the request's vector offset: the MFMA below is the wait state in front of the request
                    emit(dm[DMA_AT.index(n)][1])
                if NO_MFMA:
                    emit("s_nop 0")
                else:
                    fa, fb = frag(aset, i, kk), frag(bset, j, kk)
                    emit(f"v_mfma_f32_16x16x32_bf16 {acc(4 * ih + i, 4 * jh + j)}, v[{fb}:{fb + 3}], v[{fa}:{fa + 3}], {acc(4 * ih + i, 4 * jh + j)}")
                if n <= 16:
                    emit(rd[n - 1])
                elif n in DMA_AT:
                    emit(dm[DMA_AT.index(n)][2])
    for line in post:
        emit(line)


def ktile(stage, bc, bn):
    # requests (K tile t + 2, same stage): P1 A h0, P2 B h0 (both read in P3 / P4 of the previous K tile), P3 B h1, P4 A h1 (read in P1 / P2)
    st, so = stage, stage ^ 1
    phase(0, 0, "Ax", bc, ("B", bn, st, 1), ("A", st, 0))
    phase(0, 1, "Ax", bn, ("A", "Ay", st, 1), ("B", st, 0), barrier=False)
    phase(1, 1, "Ay", bn, ("A", "Ax", so, 0), ("B", st, 1))
    phase(1, 0, "Ay", bc, ("B", bn, so, 0), ("A", st, 1), advance("A") + advance("B"), barrier=False)


def request_group(kind, stage, h):
    for j in range(4):
        for line in dma(kind, stage, h, j):
            emit(line)


# ---------------------------------------------------------------- entry: windows, scalar offsets, fragment read addresses
emit("s_mov_b32 s88, %[alo]"); emit("s_mov_b32 s89, %[ahi]"); emit("s_mov_b32 s90, %[blo]"); emit("s_mov_b32 s91, %[bhi]")
emit("s_mov_b32 s84, %[rema]"); emit("s_mov_b32 s85, %[remb]")
emit("s_mov_b32 s60, s88"); emit("s_and_b32 s61, s89, 0xffff"); emit("s_mov_b32 s62, s84"); emit("s_mov_b32 s63, 0x00020000")
emit("s_mov_b32 s64, s90"); emit("s_and_b32 s65, s91, 0xffff"); emit("s_mov_b32 s66, s85"); emit("s_mov_b32 s67, 0x00020000")
for h in range(2):
    for j in range(4):
        # 4 j k rows of the operand + the half's 64 columns (128 bytes)
        emit(f"s_mul_i32 s{68 + 4 * h + j}, %[unita], {j}"); emit(f"s_add_u32 s{68 + 4 * h + j}, s{68 + 4 * h + j}, {128 * h}")
        emit(f"s_mul_i32 s{76 + 4 * h + j}, %[unitb], {j}"); emit(f"s_add_u32 s{76 + 4 * h + j}, s{76 + 4 * h + j}, {128 * h}")
for b in range(4):
    # block b of the wave's four: slot (4 wr + b) ^ fK = (xa ^ b), 32 bytes per slot; stage 1 is 64 KB on
    emit(f"v_xor_b32 v{104 + b}, {b}, %[xa]"); emit(f"v_lshl_add_u32 v{104 + b}, v{104 + b}, 5, %[rowb]"); emit(f"v_add_u32 v{108 + b}, 0x10000, v{104 + b}")
    emit(f"v_xor_b32 v{112 + b}, {b}, %[xb]"); emit(f"v_lshl_add_u32 v{112 + b}, v{112 + b}, 5, %[rowb]"); emit(f"v_add_u32 v{116 + b}, 0x10000, v{112 + b}")
for r in range(256):
    emit(f"v_accvgpr_write_b32 a{r}, 0")
# ---------------------------------------------------------------- prologue: K tiles 0 and 1 in the order their regions are first read
emit("s_nop 4")
for stage in range(2):
    request_group("A", stage, 0); request_group("B", stage, 0); request_group("B", stage, 1); request_group("A", stage, 1)
    for line in advance("A") + advance("B"):
        emit(line)
emit("s_waitcnt vmcnt(24)")                 # A h0, B h0 of K tile 0
emit("s_barrier")
for line in reads("A", "Ax", 0, 0) + reads("B", "Bx", 0, 0):
    emit(line)
emit("s_mov_b32 s86, %[trips]")             # pairs of K tiles (launcher: >= 1)
emit("1:")
ktile(0, "Bx", "By")
ktile(1, "By", "Bx")
emit("s_sub_u32 s86, s86, 1")
emit("s_cmp_lg_u32 s86, 0")
emit("s_cbranch_scc1 1b")
# ---------------------------------------------------------------- epilogue: four passes of 32 rows through the wave's LDS patch, atomics of 256 contiguous bytes
emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
emit("s_barrier")                           # nobody reads the operand stages any more, nothing is in flight into them
emit("s_mov_b32 s92, 0")                    # row offset (bytes) of the pass's first row inside the wave's block
for q in range(4):
    for ii in range(2):
        for j in range(8):
            t = 128 + ((ii * 8 + j) % 8) * 4
            r = (8 * (2 * q + ii) + j) * 4
            for k in range(4):
                emit(f"v_accvgpr_read_b32 v{t + k}, a{r + k}")
            emit(f"ds_write_b128 %[cswr], v[{t}:{t + 3}] offset:{((ii * 16) * 132 + j * 16) * 4}")
    emit("s_waitcnt lgkmcnt(0)")
    for g in range(4):                      # 8 rows per group: 16 reads, then 8 + 8 atomics
        for r8 in range(8):
            r = g * 8 + r8
            emit(f"ds_read_b32 v{160 + r8}, %[csrd] offset:{r * 132 * 4}")
            emit(f"ds_read_b32 v{168 + r8}, %[csrd] offset:{(r * 132 + 64) * 4}")
        emit("s_waitcnt lgkmcnt(0)")
        emit("s_mov_b64 exec, %[mask0]")
        for r8 in range(8):
            emit(f"s_mul_i32 s93, %[ldc4], {g * 8 + r8}")
            emit("s_add_u32 s93, s93, s92")
            emit(f"buffer_atomic_add_f32 v{160 + r8}, %[ccol], %[cdesc], s93 offen")
        emit("s_mov_b64 exec, %[mask1]")
        for r8 in range(8):
            emit(f"s_mul_i32 s93, %[ldc4], {g * 8 + r8}")
            emit("s_add_u32 s93, s93, s92")
            emit(f"buffer_atomic_add_f32 v{168 + r8}, %[ccol], %[cdesc], s93 offen offset:256")
        emit("s_mov_b64 exec, -1")
    emit("s_mul_i32 s93, %[ldc4], 32")
    emit("s_add_u32 s92, s92, s93")
emit("s_waitcnt vmcnt(0)")

_args = [a for a in sys.argv[1:] if not a.startswith("--")]
here = _args[0] if _args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "recommendersystem_amd", "csrc")
with open(os.path.join(here, "gemm4k_asm.inc"), "w") as f:
    f.write("// generated by tools/micro/gen_gemm4k_asm.py -- do not edit\n")
    for l in out:
        f.write(f'"{l}\\n\\t"\n')
with open(os.path.join(here, "gemm4k_clobbers.inc"), "w") as f:
    f.write("// generated by tools/micro/gen_gemm4k_asm.py -- do not edit\n")
    regs = [f"v{r}" for r in range(96, 120)] + [f"v{r}" for r in range(128, 256)] + [f"a{r}" for r in range(256)] + [f"s{r}" for r in range(60, 94)]
    f.write(", ".join(f'"{r}"' for r in regs) + "\n")
print(len(out), "asm lines")
