"""Generator of the PERSISTENT form of the hand-placed K loop (the one-tile form is gen_gemm4a_asm.py): writes ONE
inline-asm statement = one 256 x 256 output tile of a workgroup's run -- K loop, the request of the NEXT tile's first two K tiles, the
bf16 store -- which the kernel executes once per tile of its run; the tile walk itself is plain scalar C++ between two executions.

What crosses a statement boundary is LDS contents and requests in flight, never registers:
  * on entry the tile's K tiles 0 and 1 are in flight (requested by the previous execution, or by this one when flags bit 0 says "first"),
    followed -- in issue order, which is retirement order -- by the 64 stores of the previous tile.  vmcnt has 6 bits: once the last store
    has been issued at most 63 operations are outstanding, so the 32 requests in front of the stores have all landed (the explicit
    s_waitcnt vmcnt(63) says the same).
  * K tile 0 and the first half of K tile 1 run WITHOUT counted waits (everything they read is prologue data): the stores get six phases
    (~1.3 us) to drain before the first wait that covers them (K tile 1, P3: A h0 / B h0 of K tile 2).
  * behind the last K tile: barrier (every wave is done with both LDS stages), the next tile's 32 requests (flags bit 1), then the stores.

Registers: a[0:255] accumulators (block (i, j) of the wave's 128 x 128 at a[(8 i + j) 4 ..+3]); v[128:255] four fragment sets A_x, A_y, B_x,
B_y (fragment (b, kk) at set + (2 b + kk) 4), re-used as conversion temporaries by the store; v[120:123] stage-1 read addresses;
s[60:67] the two operand descriptors, s[68:83] scalar row offsets, s[84:87] counters, s[88:91] the running operand windows.
    python tools/micro/gen_gemm4p_asm.py [flags] [outdir]   (writes gemm4p_asm.inc and gemm4p_clobbers.inc; default outdir: recommendersystem_amd/csrc,
    whose copies are committed -- tests/test_host_logic.py regenerates and compares them; flags: timing-only / measurement variants, below)
"""
import os
import sys

NO_MFMA = "--no-mfma" in sys.argv     # timing-only variant (wrong results): the request / wait / barrier skeleton of the loop alone
A_EMPTY = "--a-empty" in sys.argv     # timing-only: every A (B) request goes through an empty window -- counted like a real one, fetches nothing
B_EMPTY = "--b-empty" in sys.argv
PF = "--pf" in sys.argv               # every even K tile also touches, per A row, the 128 bytes BEHIND the ones it requests (one dword per lane into a
                                      # scratch register): the odd K tile's request then hits the L2, and the memory sees 256 contiguous bytes per row and visit
DEEP = "--deep" in sys.argv           # a barrier and a counted wait in EVERY phase, each region re-requested one phase after its last read: 24 requests
                                      # (96 KB per CU) in flight at the waits instead of 16 (64 KB), six phases for a request to land instead of five
A_PACKED = "--a-packed" in sys.argv   # timing-only (with -DGEMM4P_A_PACKED): A read as if stored tile by tile -- a K tile of 256 rows = 32 KB contiguous
DMA_AT = (11, 15, 19, 23)             # behind which MFMAs of a phase (1 .. 32) its four requests are issued; --dma-at=a,b,c,d: measurement
for _a in sys.argv[1:]:
    if _a.startswith("--dma-at="):
        DMA_AT = tuple(int(x) for x in _a.split("=")[1].split(","))
NT_STORE = "--nt-store" in sys.argv   # measurement: the output tile stored with the non-temporal policy (what the vendor's kernel for this class does: "NTD")
SET = {"Ax": 128, "Ay": 160, "Bx": 192, "By": 224}
out = []


def emit(s):
    out.append(s)


def frag(setname, b, kk):
    r = SET[setname] + (2 * b + kk) * 4
    return f"v[{r}:{r + 3}]"


def acc(i, j):
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
    for b in range(4):
        for kk in range(2):
            addr = f"%[r{kind.lower()}{kk}]" if stage == 0 else f"v{120 + (0 if kind == 'A' else 2) + kk}"
            res.append(f"ds_read_b128 {frag(setname, b, kk)}, {addr} offset:{h * 16384 + b * 2048}")
    return res


def advance(kind, count=True):
    """window of `kind` one K tile on; with `count` also the remaining-K-tiles counter and the empty window behind the last K tile"""
    lo, hi, rem, recw, full = ("s88", "s89", "s84", "s62", "0" if A_EMPTY else "%[reca]") if kind == "A" else ("s90", "s91", "s85", "s66", "0" if B_EMPTY else "%[recb]")
    d0, d1 = ("s60", "s61") if kind == "A" else ("s64", "s65")
    res = [f"s_add_u32 {lo}, {lo}, {32768 if (A_PACKED and kind == 'A') else 128}", f"s_addc_u32 {hi}, {hi}, 0", f"s_mov_b32 {d0}, {lo}", f"s_and_b32 {d1}, {hi}, 0xffff"]
    if count:
        res += [f"s_sub_i32 {rem}, {rem}, 1", f"s_cmp_gt_i32 {rem}, 0", f"s_cselect_b32 {recw}, {full}, 0"]
    return res


def phase(ih, jh, aset, bset, pref, dmas, post=(), barrier=True, wait=16, pf=False):
    """pref = (kind, set, stage, h) fragment reads for the next phase; dmas = (kind, stage, h) region to request; post: scalar bookkeeping"""
    emit("s_waitcnt lgkmcnt(0)")
    if barrier:
        if wait:
            emit(f"s_waitcnt vmcnt({wait})")
        emit("s_barrier")
    rd = reads(*pref)
    dm = [dma(dmas[0], dmas[1], dmas[2], j) for j in range(4)]
    n = 0
    for i in range(4):
        for j in range(4):
            for kk in range(2):
                n += 1
                if n in DMA_AT:
                    emit(dm[DMA_AT.index(n)][0])        # s_add_u32 m0: the MFMA below is its wait state in front of the DMA
                if NO_MFMA:
                    emit("s_nop 0")
                else:
                    emit(f"v_mfma_f32_16x16x32_bf16 {acc(4 * ih + i, 4 * jh + j)}, {frag(bset, j, kk)}, {frag(aset, i, kk)}, {acc(4 * ih + i, 4 * jh + j)}")
                if n <= 8:
                    emit(rd[n - 1])
                elif n in DMA_AT:
                    emit(dm[DMA_AT.index(n)][2])
                elif n == 26 and pf:
                    emit("buffer_load_dword v119, %[vpf], s[60:63], 0 offen offset:128")
    for line in post:
        emit(line)


def ktile(stage, bc, bn, wait13=(16, 16), pf=False, deep_waits=(24, 24, 24, 24)):
    st, so = stage, stage ^ 1
    if DEEP:
        # requests: P1 B h0 (t + 2; read in P4 of the previous K tile), P2 B h1 (t + 2; read in P1), P3 A h1 (t + 2; read in P2), P4 A h0 of the
        # OTHER stage (t + 3; read in P3); the B window moves on behind P2, the A window behind P3
        phase(0, 0, "Ax", bc, ("B", bn, st, 1), ("B", st, 0), wait=deep_waits[0])
        phase(0, 1, "Ax", bn, ("A", "Ay", st, 1), ("B", st, 1), advance("B"), wait=deep_waits[1])
        phase(1, 1, "Ay", bn, ("A", "Ax", so, 0), ("A", st, 1), advance("A"), wait=deep_waits[2])
        phase(1, 0, "Ay", bc, ("B", bn, so, 0), ("A", so, 0), wait=deep_waits[3])
        return
    # requests (K tile t + 2, same stage): P1 A h0, P2 B h0 (both read in P3 / P4 of the previous K tile), P3 B h1, P4 A h1 (read in P1 / P2)
    phase(0, 0, "Ax", bc, ("B", bn, st, 1), ("A", st, 0), wait=wait13[0], pf=pf)
    phase(0, 1, "Ax", bn, ("A", "Ay", st, 1), ("B", st, 0), barrier=False)
    phase(1, 1, "Ay", bn, ("A", "Ax", so, 0), ("B", st, 1), wait=wait13[1])
    phase(1, 0, "Ay", bc, ("B", bn, so, 0), ("A", st, 1), advance("A") + advance("B"), barrier=False)


def request_group(kind, stage, h):
    for j in range(4):
        for line in dma(kind, stage, h, j):
            emit(line)


def set_windows(alo, ahi, blo, bhi):
    emit(f"s_mov_b32 s88, {alo}"); emit(f"s_mov_b32 s89, {ahi}"); emit(f"s_mov_b32 s90, {blo}"); emit(f"s_mov_b32 s91, {bhi}")
    emit("s_mov_b32 s60, s88"); emit("s_and_b32 s61, s89, 0xffff"); emit("s_mov_b32 s62, " + ("0" if A_EMPTY else "%[reca]")); emit("s_mov_b32 s63, 0x00020000")
    emit("s_mov_b32 s64, s90"); emit("s_and_b32 s65, s91, 0xffff"); emit("s_mov_b32 s66, " + ("0" if B_EMPTY else "%[recb]")); emit("s_mov_b32 s67, 0x00020000")


def prologue_requests():
    """K tiles 0 and 1 of the tile the windows point at, in the order their regions are first read; leaves the windows at K tile 2"""
    for stage in range(2):
        request_group("A", stage, 0); request_group("B", stage, 0); request_group("B", stage, 1); request_group("A", stage, 1)
        for line in advance("A", False) + advance("B", False):
            emit(line)


# ---------------------------------------------------------------- entry: windows of this tile, scalar offsets, stage-1 read addresses
set_windows("%[alo]", "%[ahi]", "%[blo]", "%[bhi]")
for h in range(2):
    for j in range(4):
        emit(f"s_mul_i32 s{68 + 4 * h + j}, %[unita], {8 * h + j}")     # (h * 64 + j * 8) rows * lda * 2 bytes
        emit(f"s_mul_i32 s{76 + 4 * h + j}, %[unitb], {8 * h + j}")
emit("v_add_u32 v120, 0x10000, %[ra0]"); emit("v_add_u32 v121, 0x10000, %[ra1]"); emit("v_add_u32 v122, 0x10000, %[rb0]"); emit("v_add_u32 v123, 0x10000, %[rb1]")
emit("s_bitcmp1_b32 %[flags], 0")
emit("s_cbranch_scc0 2f")
# first tile of the run: its own prologue
prologue_requests()
emit("s_waitcnt vmcnt(0)")                  # (K tiles 0 and 1 run without counted waits: everything has to be there; once per run)
emit("s_branch 3f")
emit("2:")
# K tiles 0 and 1 were requested by the previous execution: the windows start at K tile 2
for _ in range(2):
    for line in advance("A", False) + advance("B", False):
        emit(line)
emit("s_waitcnt vmcnt(63)")                 # (see the header: implied by the 64 stores issued behind the 32 requests)
emit("3:")
# remaining valid K tiles from K tile 2 on
emit("s_sub_i32 s84, %[nt], 2"); emit("s_cmp_gt_i32 s84, 0"); emit("s_cselect_b32 s62, " + ("0" if A_EMPTY else "%[reca]") + ", 0")
emit("s_sub_i32 s85, %[nt], 2"); emit("s_cmp_gt_i32 s85, 0"); emit("s_cselect_b32 s66, " + ("0" if B_EMPTY else "%[recb]") + ", 0")
for r in range(256):
    emit(f"v_accvgpr_write_b32 a{r}, 0")
emit("s_barrier")
for line in reads("A", "Ax", 0, 0) + reads("B", "Bx", 0, 0):
    emit(line)
# ---------------------------------------------------------------- K loop: K tiles 0, 1 on prologue data, then pairs
if DEEP:
    # the request the fourth phase of "K tile -1" would have made: A h0 of K tile 2 into the region just read.  The first wait that guards a
    # request issued behind the previous tile's stores is P3 of K tile 1 (it reads this one), with the steady state's 24 younger ones
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    request_group("A", 0, 0)
    ktile(0, "Bx", "By", deep_waits=(0, 0, 0, 0))
    ktile(1, "By", "Bx", deep_waits=(0, 0, 24, 24))
else:
    ktile(0, "Bx", "By", (0, 0), pf=PF)
    ktile(1, "By", "Bx", (0, 16))
emit("s_lshr_b32 s86, %[nt], 1")
emit("s_sub_u32 s86, s86, 1")               # launcher: nt even, >= 4
emit("1:")
if PF:      # (the touch is issued behind P1's four requests: it is among the youngest 17 at the next two waits)
    ktile(0, "Bx", "By", (16, 17), pf=True)
    ktile(1, "By", "Bx", (17, 16))
else:
    ktile(0, "Bx", "By")
    ktile(1, "By", "Bx")
emit("s_sub_u32 s86, s86, 1")
emit("s_cmp_lg_u32 s86, 0")
emit("s_cbranch_scc1 1b")
# ---------------------------------------------------------------- between two tiles
emit("s_waitcnt lgkmcnt(0)")                # the fragment reads of the phase behind the last one (nobody uses them) have returned
emit("s_barrier")                           # every wave is done with both stages
emit("s_bitcmp1_b32 %[flags], 1")
emit("s_cbranch_scc0 4f")
set_windows("%[nalo]", "%[nahi]", "%[nblo]", "%[nbhi]")
prologue_requests()
emit("4:")
# ---------------------------------------------------------------- bf16 store of the 8 x 8 blocks: row offsets in s[68:75] (re-used), columns as immediates
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
        emit(f"buffer_store_dwordx2 v[{t + 4}:{t + 5}], %[vc], %[cdesc], s{68 + i} offen offset:{j * 32}" + (" nt" if NT_STORE else ""))
emit("s_bitcmp1_b32 %[flags], 1")
emit("s_cbranch_scc1 5f")
emit("s_waitcnt vmcnt(0)")                  # last tile of the run
emit("5:")

_args = [a for a in sys.argv[1:] if not a.startswith("--")]
here = _args[0] if _args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "recommendersystem_amd", "csrc")
with open(os.path.join(here, "gemm4p_asm.inc"), "w") as f:
    f.write("// generated by tools/micro/gen_gemm4p_asm.py -- do not edit\n")
    for l in out:
        f.write(f'"{l}\\n\\t"\n')
with open(os.path.join(here, "gemm4p_clobbers.inc"), "w") as f:
    f.write("// generated by tools/micro/gen_gemm4p_asm.py -- do not edit\n")
    regs = [f"v{r}" for r in range(119, 124)] + [f"v{r}" for r in range(128, 256)] + [f"a{r}" for r in range(256)] + [f"s{r}" for r in range(60, 92)]
    f.write(", ".join(f'"{r}"' for r in regs) + "\n")
print(len(out), "asm lines")
