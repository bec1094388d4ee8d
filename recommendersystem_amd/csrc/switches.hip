// switches.hpp: the one place that reads RSYS_* from the environment.
#include "switches.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace rsys {
namespace {

struct Entry { const char* name; int Switches::*field; int dflt; };
const Entry kEntries[] = {
    {"RSYS_ATTN_DMA", &Switches::attn_dma, 1},
    {"RSYS_ATTN_KV_DMA", &Switches::attn_kv_dma, 1},
    {"RSYS_ATTN_KV32", &Switches::attn_kv32, 1},
    {"RSYS_ATTN_FWD32", &Switches::attn_fwd32, 0},
    {"RSYS_GEMM_KERNEL", &Switches::gemm_kernel, -1},
    {"RSYS_GEMM_KERNEL_TN", &Switches::gemm_kernel_tn, -1},
    {"RSYS_GEMM_KERNEL_NT_SPLITK", &Switches::gemm_kernel_nt_splitk, -1},
    {"RSYS_GEMM_KERNEL_MIX", &Switches::gemm_kernel_mix, 1},
    {"RSYS_GEMM8C_HALF", &Switches::gemm8c_half, 1},
    {"RSYS_GEMM4P", &Switches::gemm4p, 1},
    {"RSYS_GEMM4K", &Switches::gemm4k, 1},
    {"RSYS_GEMM8C", &Switches::gemm8c, 1},
    {"RSYS_GEMM_REVERSE", &Switches::gemm_reverse, 1},
    {"RSYS_GEMM_PATCH", &Switches::gemm_patch, 1},
    {"RSYS_TABLE_TAIL", &Switches::table_tail, 1},
    {"RSYS_DW_GROUP", &Switches::dw_group, 1},
    {"RSYS_DET_DW_GROUP", &Switches::det_dw_group, 1},
    {"RSYS_SPARSE_TOP", &Switches::sparse_top, 1},
    {"RSYS_TOP_ORDER", &Switches::top_order, 1},
    {"RSYS_SELECT_ASIDE", &Switches::select_aside, 0},
    {"RSYS_SELECT_CHUNKED", &Switches::select_chunked, 1},
    {"RSYS_SIDE_STREAM", &Switches::side_stream, 0},
    {"RSYS_SCATTER_ATOMIC", &Switches::scatter_atomic, 0},
    {"RSYS_F8_DW", &Switches::f8_dw, 1},
    {"RSYS_F8_DW_ROUND_BF16", &Switches::f8_dw_round_bf16, 0},
    {"RSYS_F8_DEBUG_KEEP", &Switches::f8_debug_keep, 0},
    {"RSYS_FORCE_RCCL", &Switches::force_rccl, 0},
    {"RSYS_DEBUG_8P", &Switches::debug_8p, 0},
    {"RSYS_DEBUG_8G_SPLITK", &Switches::debug_8g_splitk, 0},
    {"RSYS_DEBUG_8T_SPLITK", &Switches::debug_8t_splitk, 0},
    {"RSYS_DEBUG_EPI", &Switches::debug_epi, -1},
    {"RSYS_DEBUG_F8_CAST_WAVES", &Switches::debug_f8_cast_waves, 4},
    {"RSYS_DEBUG_NORM_BWD_GRID", &Switches::debug_norm_bwd_grid, 1024},
    {"RSYS_DEBUG_NORM_BWD_WAVES", &Switches::debug_norm_bwd_waves, 0},
    {"RSYS_DEBUG_ADAMW", &Switches::debug_adamw, 0},
};

Switches g_sw;
bool g_parsed = false;
std::mutex g_mu;

void parse_locked() {
  for (const Entry& e : kEntries) {
    const char* v = getenv(e.name);
    g_sw.*(e.field) = (v && *v) ? atoi(v) : e.dflt;
  }
  g_parsed = true;
}

}  // namespace

void switches_parse() { std::lock_guard<std::mutex> lk(g_mu); parse_locked(); }

const Switches& sw() {
  if (!g_parsed) { std::lock_guard<std::mutex> lk(g_mu); if (!g_parsed) parse_locked(); }
  return g_sw;
}

int switches_describe(char* buf, int cap) {
  const Switches& s = sw();
  int need = 0;
  if (buf && cap > 0) buf[0] = 0;
  for (const Entry& e : kEntries) {
    if (s.*(e.field) == e.dflt) continue;
    char one[96];
    const int n = snprintf(one, sizeof one, "%s%s=%d", need ? " " : "", e.name, s.*(e.field));
    if (buf && need + n < cap) memcpy(buf + need, one, (size_t)n + 1);
    need += n;
  }
  return need;
}

}  // namespace rsys
