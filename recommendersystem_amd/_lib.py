"""ctypes binding of librsys_hip.so (C ABI: include/rsys.h; test and parity hooks: include/rsys_debug.h).

The HIP library is the product path: there is no CPU fallback.  Importing this
module without the built library raises; calling compute entry points without a
GPU returns an error from the library ("no HIP device visible").
"""
import ctypes as C
import os

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL's ncclCommInitRank fails with
# "hipIpcGetMemHandle: invalid argument" otherwise); set before the HIP runtime is first loaded, never overriding the caller
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librsys_hip.so")


class RsysError(RuntimeError):
    pass


class rsys_config(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
        ("embed_dim", C.c_int32), ("intermediate_dim", C.c_int32), ("max_sequence_length", C.c_int32),
        ("vocab_0", C.c_int32), ("vocab_1", C.c_int32),
        ("vocab_status", C.c_int32), ("vocab_gender", C.c_int32), ("vocab_source", C.c_int32),
        ("metadata_dim", C.c_int32),
        ("min_ts", C.c_double), ("max_ts", C.c_double),
        ("rating_mean", C.c_float), ("rating_std", C.c_float), ("mask_rate", C.c_float),
        ("mask_topk", C.c_int32), ("finetune", C.c_int32), ("finetune_metric", C.c_int32),
        ("dtype", C.c_int32), ("max_rows", C.c_int32), ("lora_dropout", C.c_float),
        ("table_shard_rank", C.c_int32), ("table_shard_world", C.c_int32), ("sampled_negatives", C.c_int32),
    ]


class rsys_batch(C.Structure):
    _fields_ = [
        ("rows", C.c_int32),
        ("userid", C.c_void_p), ("token_mask_ids", C.c_void_p), ("gender", C.c_void_p), ("source", C.c_void_p),
        ("matchedid", C.c_void_p), ("status", C.c_void_p),
        ("time", C.c_void_p), ("rating", C.c_void_p), ("progress", C.c_void_p),
        ("label", C.c_void_p * 6), ("weight", C.c_void_p * 6), ("position", C.c_void_p * 6),
        ("watch_mask", C.c_void_p), ("rating_mask", C.c_void_p), ("rope_input_pos", C.c_void_p),
    ]


# every symbol include/rsys.h and include/rsys_debug.h declare: (name, restype, argtypes)
_P = C.c_void_p
_SIGS = [
    ("rsys_version", C.c_char_p, []),
    ("rsys_last_error", C.c_size_t, [C.c_char_p, C.c_size_t]),
    ("rsys_device_count", C.c_int32, [C.POINTER(C.c_int32)]),
    ("rsys_device_synchronize", C.c_int32, []),
    ("rsys_model_create", C.c_int32, [C.POINTER(rsys_config), C.c_int32, C.POINTER(_P)]),
    ("rsys_model_destroy", C.c_int32, [_P]),
    ("rsys_model_init_random", C.c_int32, [_P, C.c_uint64]),
    ("rsys_model_load_metadata", C.c_int32, [_P, _P, C.c_int64, C.c_int64]),
    ("rsys_model_random_metadata", C.c_int32, [_P, C.c_uint64]),
    ("rsys_model_set_rope", C.c_int32, [_P, _P, _P, C.c_int64]),
    ("rsys_param_count", C.c_int32, [_P, C.POINTER(C.c_int32)]),
    ("rsys_param_info", C.c_int32, [_P, C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64 * 2), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("rsys_param_get", C.c_int32, [_P, C.c_char_p, _P, C.c_int64]),
    ("rsys_param_set", C.c_int32, [_P, C.c_char_p, _P, C.c_int64]),
    ("rsys_grad_get", C.c_int32, [_P, C.c_char_p, _P, C.c_int64]),
    ("rsys_zero_grad", C.c_int32, [_P]),
    ("rsys_batch_upload", C.c_int32, [_P, C.POINTER(rsys_batch)]),
    ("rsys_batch_prefetch", C.c_int32, [_P, C.POINTER(rsys_batch)]),
    ("rsys_batch_swap", C.c_int32, [_P]),
    ("rsys_forward_backward", C.c_int32, [_P, C.c_int32, C.POINTER(C.c_float * 4), C.c_float, C.c_uint64, C.c_uint64]),
    ("rsys_losses_get", C.c_int32, [_P, C.POINTER(C.c_float * 12), C.POINTER(C.c_float * 4)]),
    ("rsys_losses_push", C.c_int32, [_P]),
    ("rsys_losses_drain", C.c_int32, [_P, _P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    ("rsys_head_rows_get", C.c_int32, [_P, C.POINTER(C.c_int32 * 4)]),
    ("rsys_item_table", C.c_int32, [_P, _P, C.c_int64]),
    ("rsys_model_set_deterministic", C.c_int32, [_P, C.c_int32]),
    ("rsys_infer", C.c_int32, [_P, C.c_int32, _P, C.c_int64]),
    ("rsys_infer_select", C.c_int32, [_P, C.c_int32, _P, C.c_int64, _P, C.c_int64]),
    ("rsys_trunk_output_get", C.c_int32, [_P, _P, C.c_int64]),
    ("rsys_debug_get", C.c_int32, [_P, C.c_char_p, _P, C.c_int64]),
    ("rsys_clip_grad_norm", C.c_int32, [_P, C.c_float, C.POINTER(C.c_float)]),
    ("rsys_adamw_create", C.c_int32, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(_P)]),
    ("rsys_adamw_destroy", C.c_int32, [_P]),
    ("rsys_adamw_step", C.c_int32, [_P, C.c_float, C.c_float, C.c_float]),
    ("rsys_adamw_set_zero1", C.c_int32, [_P, C.c_int32, C.c_int32]),
    ("rsys_adamw_step_zero1", C.c_int32, [_P, _P, C.c_float, C.c_float, C.c_float]),
    ("rsys_adamw_state_get", C.c_int32, [_P, C.c_char_p, _P, _P, C.c_int64, C.POINTER(C.c_int32)]),
    ("rsys_adamw_state_set", C.c_int32, [_P, C.c_char_p, _P, _P, C.c_int64, C.c_int32]),
    ("rsys_comm_unique_id", C.c_int32, [C.POINTER(C.c_uint8 * 128)]),
    ("rsys_comm_init", C.c_int32, [C.POINTER(C.c_uint8 * 128), C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("rsys_comm_destroy", C.c_int32, [_P]),
    ("rsys_comm_debug_delay", C.c_int32, [_P, C.c_int32]),
    ("rsys_set_grad_sync", C.c_int32, [_P, _P]),
    ("rsys_model_set_split_table_reduce", C.c_int32, [_P, C.c_int32]),
    ("rsys_model_set_shard_comm", C.c_int32, [_P, _P]),
    ("rsys_table_rows", C.c_int32, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("rsys_local_group_create", C.c_int32, [C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("rsys_local_group_destroy", C.c_int32, [_P]),
    ("rsys_comm_init_local", C.c_int32, [_P, C.c_int32, C.POINTER(_P)]),
    ("rsys_allreduce_grads", C.c_int32, [_P, _P]),
    ("rsys_grad_sync_early", C.c_int32, [_P, C.POINTER(C.c_int64)]),
    ("rsys_grad_sync_schedule", C.c_int32, [_P, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32)]),
    ("rsys_comm_info", C.c_int32, [_P, C.POINTER(C.c_int32)]),
    ("rsys_allreduce_f64", C.c_int32, [_P, C.POINTER(C.c_double), C.c_int32]),
    ("rsys_self_test", C.c_int32, [_P]),
    ("rsys_param_checksum", C.c_int32, [_P, C.POINTER(C.c_double * 4)]),
    ("rsys_grad_buffer", C.c_int32, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    ("rsys_param_buffer", C.c_int32, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    ("rsys_refresh_shadow", C.c_int32, [_P]),
    ("rsys_dev_alloc", C.c_int32, [C.POINTER(_P), C.c_size_t]),
    ("rsys_dev_free", C.c_int32, [_P]),
    ("rsys_dev_h2d", C.c_int32, [_P, _P, C.c_size_t]),
    ("rsys_dev_d2h", C.c_int32, [_P, _P, C.c_size_t]),
    ("rsys_dev_memset", C.c_int32, [_P, C.c_int, C.c_size_t]),
    ("rsys_op_gemm", C.c_int32, [C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                 C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("rsys_op_f8_quantize", C.c_int32, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P, _P, _P,
                                          C.c_int32, C.c_int32, C.c_int32]),
    ("rsys_op_f8_weights", C.c_int32, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64]),
    ("rsys_op_gemm_f8", C.c_int32, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("rsys_op_gemm_rows", C.c_int32, [C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                      C.c_int32, C.c_int32, _P]),
    ("rsys_op_gemm_klimit", C.c_int32, [C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _P]),
    ("rsys_op_attention", C.c_int32, [C.c_int32] + [C.c_int32] * 5 + [_P] * 9),
    ("rsys_op_embedding_scatter", C.c_int32, [_P, C.c_int64, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32]),
    ("rsys_step_mark", C.c_int32, [_P]),
    ("rsys_step_marks_get", C.c_int32, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    ("rsys_op_timing", C.c_int32, [_P, C.c_int32]),
    ("rsys_timing_get", C.c_int32, [_P, C.c_char_p, C.c_size_t]),
    ("rsys_op_timing_filter", C.c_int32, [_P, C.c_char_p]),
    ("rsys_switches_reload", C.c_int32, []),
    ("rsys_switches_describe", C.c_int32, [C.c_char_p, C.c_int32]),
]
EXPORTED = [s[0] for s in _SIGS]

_lib = None


def lib():
    """The loaded library; raises if it has not been built (python __graft_entry__.py / make)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsysError(f"{LIB_PATH} is missing: build it with `make -C recommendersystem_amd/csrc` "
                            "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, res, args in _SIGS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(2048)
    lib().rsys_last_error(buf, 2048)
    return buf.value.decode(errors="replace")


def check(rc):
    if rc != 0:
        raise RsysError(f"rsys error {rc}: {last_error()}")


def device_count():
    n = C.c_int32(0)
    check(lib().rsys_device_count(C.byref(n)))
    return n.value


def switches():
    """The RSYS_* environment switches that differ from their defaults, as {name: value} (csrc/switches.hpp; parsed now)."""
    L = lib()
    L.rsys_switches_reload()
    buf = C.create_string_buffer(4096)
    L.rsys_switches_describe(buf, len(buf))
    out = {}
    for item in buf.value.decode().split():
        k, v = item.split("=")
        out[k] = int(v)
    return out
