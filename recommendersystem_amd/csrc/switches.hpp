// Every environment switch of the library in one place.  The environment is parsed by switches_parse() and nowhere
// else: at the first use, at rsys_model_create / rsys_comm_* creation, and at the entry of the standalone rsys_op_* operators (the
// entry points the per-kernel tests and tools/bench_*.py call: they flip a switch between two calls of one process).  A training step
// reads the parsed struct only.  DESIGN.md "Environment switches" lists each field with its default, what the other arm is kept for
// and the test that runs it; rsys_switches_describe() prints the values that differ from the defaults (bench.py puts them in its line).
#pragma once

namespace rsys {

struct Switches {
  // ---- kernel choice: both arms live code (the other arm serves other shapes / dtypes, or is the reference side of a bitwise test)
  int attn_dma;               // RSYS_ATTN_DMA=0: bf16 / head_dim 64 on the register-staged attention kernels (every other head size's kernels)
  int attn_kv_dma;            // RSYS_ATTN_KV_DMA=0: dK/dV alone on the register-staged kernel
  int attn_fwd32;             // RSYS_ATTN_FWD32=1: forward on attn_fwd32_kernel (128 queries of one head per workgroup, 32x32x16 products; measured level with the default 64-query two-head kernel)
  int attn_kv32;              // RSYS_ATTN_KV32=0: dK/dV on the 16-key-per-wave LDS-DMA kernel (the bitwise partner of the register-staged one)
  int gemm_kernel;            // RSYS_GEMM_KERNEL: -1 unset / 0 (shape rule); 1 = the 128x128 register-staged kernel everywhere, 2 = 256x256 LDS-DMA wherever eligible
  int gemm_kernel_tn;         // RSYS_GEMM_KERNEL_TN: -1 unset / 0 (shape rule); 1 = never the K-major LDS-DMA kernels, 2 = both their forms (split-K, store) wherever eligible
  int gemm_kernel_nt_splitk;  // RSYS_GEMM_KERNEL_NT_SPLITK: -1 unset (shape rule); 0 off; 2 force
  int gemm_kernel_mix;        // RSYS_GEMM_KERNEL_MIX: 1 = row-major A x K-major B atomic products with K >= 8192 and N >= 512 on the mixed-layout LDS-DMA kernel (gemm8p_mix_kernel); 0 = never; 2 = wherever eligible
  int gemm8c_half;            // RSYS_GEMM8C_HALF: 1 = 128x256 output tiles for outputs with fewer 256x256 tiles than CUs; 0 = never; 2 = wherever the class has the kernel
  int gemm4p;                 // RSYS_GEMM4P: 1 = plain bf16 stores with K >= 8192 on outputs >= 4 tiles wide on the four-wave register-named K loop (gemm4p.hip); 0 = never; 2 = wherever eligible
  int gemm4k;                 // RSYS_GEMM4K=0: K-major split-K products (weight gradients) on gemm8p's eight-wave kernels instead of the four-wave register-named loop (gemm4k.hip)
  int gemm8c;                 // RSYS_GEMM8C=0: row-major 256x256 products on gemm8p.hip (per-tile operand requests) instead of gemm8c.hip
  int gemm_reverse;           // RSYS_GEMM_REVERSE: 1 = the consumers of a just-written large activation (w2_fwd, w13_dx) walk their tile rows from the last to the first; 0 = never; 2 = every gemm8c launch
  int gemm_patch;             // RSYS_GEMM_PATCH: 1 = band order of the output tiles for wide and tall outputs; 0 = row-major everywhere; 2 = bands everywhere
  int table_tail;             // RSYS_TABLE_TAIL=0: the fused item table in ONE launch (its last persistent round on a fraction of the CUs) instead of main launch + split-K tail
  int dw_group;               // RSYS_DW_GROUP=0: one weight-gradient launch per product instead of the grouped launch
  int det_dw_group;           // RSYS_DET_DW_GROUP=0: deterministic mode on the per-layer slab path instead of the ordered grouped launch
  int sparse_top;             // RSYS_SPARSE_TOP=0: dense last layer and final norm
  int top_order;              // RSYS_TOP_ORDER=0: the last layer's attention keeps the token order (test hook: two summation orders of one arithmetic)
  int select_aside;           // RSYS_SELECT_ASIDE=1: position selection on the side stream
  int select_chunked;         // RSYS_SELECT_CHUNKED=0: one-pass position selection
  int side_stream;            // RSYS_SIDE_STREAM: 0 = off (default), 1 / 2 = the two side-stream placements of profiles/r4_ab_side_stream*
  int scatter_atomic;         // RSYS_SCATTER_ATOMIC=1: embedding-gradient scatter by float atomics (the A/B partner of the segmented sum)
  int f8_dw;                  // RSYS_F8_DW=0: fp8 trunk with bf16 weight gradients
  int f8_dw_round_bf16;       // RSYS_F8_DW_ROUND_BF16=1: fp8 weight-gradient operands rounded through bf16 first (parity test of the quantiser)
  int f8_debug_keep;          // RSYS_F8_DEBUG_KEEP=1: keep the fp8 backward's intermediate products for the stage-wise parity tests
  int force_rccl;             // RSYS_FORCE_RCCL=1: RCCL communicator even for in-process ranks
  // ---- measurement knobs (tools/ only; defaults are the shipped values)
  int debug_8p;               // RSYS_DEBUG_8P: GemmParams::flags bits of gemm8p / gemm8c timing experiments
  int debug_8g_splitk;        // RSYS_DEBUG_8G_SPLITK: force the grouped launch's K-split count
  int debug_8t_splitk;        // RSYS_DEBUG_8T_SPLITK: force the K-major kernel's K-split count
  int debug_epi;              // RSYS_DEBUG_EPI: rsys_op_gemm only: epilogue class override (99 = none)
  int debug_f8_cast_waves;    // RSYS_DEBUG_F8_CAST_WAVES: waves per workgroup of the fp8 cast kernel (1..4)
  int debug_norm_bwd_grid;    // RSYS_DEBUG_NORM_BWD_GRID: cap of rmsnorm_bwd's waves / 4
  int debug_adamw;            // RSYS_DEBUG_ADAMW: bit 0 = plain instead of nontemporal loads / stores in AdamW (A/B partner), bit 1 = 8192 workgroups
  int debug_norm_bwd_waves;   // RSYS_DEBUG_NORM_BWD_WAVES: waves per workgroup of rmsnorm_bwd (0 = 16 up to D = 1024, else 4)
};

const Switches& sw();        // the parsed switches (parses at first use)
void switches_parse();       // re-read the environment
// "NAME=value" of every switch that differs from its default, space separated; returns the number of characters needed
int switches_describe(char* buf, int cap);

}  // namespace rsys
