# Julia binding of librsys_hip.so (C ABI: include/rsys.h) -- thin `ccall` wrappers, 1:1 with the header: every entry point of
# the header is bound here; the test and parity hooks of include/rsys_debug.h (index-path read-back, in-process rank group, per-kernel
# access) are not part of the boundary and are not bound.  tests/test_julia_binding.py parses the `ccall` tuples and the two struct mirrors below and checks
# names, arity and C types against include/rsys.h (Julia itself cannot run in the build image).
#
# NOT EXECUTED in the build container (Julia is absent from the image, SURVEY.md 8(c)); it is the stub a
# maintainer adds so that notebooks/Training/run.jl:70 (`julia rungpu.jl`, which remote-launches torchrun
# transformer.py) can call the HIP path in-process instead.  Host arrays are kept alive with GC.@preserve
# for the duration of each call; the library owns all device memory.
module RsysHIP

const LIB = joinpath(@__DIR__, "..", "recommendersystem_amd", "librsys_hip.so")

# RsysConfig.dtype (rsys.h RSYS_DTYPE_*): fp32 parity mode, the bf16 autocast arithmetic, or bf16 with the blocks' linears on
# tensor-wise scaled fp8 operands -- what transformer.py:671-676 gets from torchao's convert_to_float8_training
const DTYPE_FP32 = Int32(0); const DTYPE_BF16 = Int32(1); const DTYPE_FP8 = Int32(2)

struct RsysConfig              # mirrors rsys_config (field order and types as in rsys.h)
    num_layers::Int32; num_heads::Int32; num_kv_heads::Int32; embed_dim::Int32; intermediate_dim::Int32
    max_sequence_length::Int32
    vocab_0::Int32; vocab_1::Int32
    vocab_status::Int32; vocab_gender::Int32; vocab_source::Int32
    metadata_dim::Int32
    min_ts::Float64; max_ts::Float64
    rating_mean::Float32; rating_std::Float32; mask_rate::Float32
    mask_topk::Int32; finetune::Int32; finetune_metric::Int32
    dtype::Int32; max_rows::Int32
    lora_dropout::Float32
    table_shard_rank::Int32; table_shard_world::Int32     # row-sharded item table (world 0 = replicated)
    sampled_negatives::Int32                               # sampled soft-max over the local classes (0 = full)
end

struct RsysBatch               # mirrors rsys_batch
    rows::Int32
    userid::Ptr{Int32}; token_mask_ids::Ptr{Int32}; gender::Ptr{Int32}; source::Ptr{Int32}
    matchedid::Ptr{Int32}; status::Ptr{Int32}
    time::Ptr{Float64}; rating::Ptr{Float32}; progress::Ptr{Float32}
    label::NTuple{6,Ptr{Float32}}; weight::NTuple{6,Ptr{Float32}}; position::NTuple{6,Ptr{Int32}}
    watch_mask::Ptr{UInt8}; rating_mask::Ptr{UInt8}; rope_input_pos::Ptr{Int32}
end

function last_error()
    buf = Vector{UInt8}(undef, 2048)
    ccall((:rsys_last_error, LIB), Csize_t, (Ptr{UInt8}, Csize_t), buf, length(buf))
    unsafe_string(pointer(buf))
end
check(rc) = rc == 0 ? nothing : error("rsys error $rc: $(last_error())")

mutable struct Model; h::Ptr{Cvoid}; end
mutable struct Optimizer; h::Ptr{Cvoid}; end
mutable struct Comm; h::Ptr{Cvoid}; world::Int; end

function Model(cfg::RsysConfig, device::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rsys_model_create, LIB), Int32, (Ref{RsysConfig}, Int32, Ref{Ptr{Cvoid}}), cfg, device, h))
    m = Model(h[]); finalizer(x -> ccall((:rsys_model_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), m); m
end
init_weights!(m::Model, seed::Integer) = check(ccall((:rsys_model_init_random, LIB), Int32, (Ptr{Cvoid}, UInt64), m.h, seed))
function load_pretrained_embeddings!(m::Model, W::Matrix{Float32})   # W is (M, V) column-major = (V, M) row-major (transformer.jl:58)
    GC.@preserve W check(ccall((:rsys_model_load_metadata, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64),
                               m.h, W, size(W, 2), size(W, 1)))
end
function set_parameter!(m::Model, name::String, x::Array{Float32})
    GC.@preserve x check(ccall((:rsys_param_set, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Float32}, Int64), m.h, name, x, length(x)))
end
function get_parameter!(m::Model, name::String, out::Array{Float32})
    GC.@preserve out check(ccall((:rsys_param_get, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Float32}, Int64), m.h, name, out, length(out)))
    out
end
function upload!(m::Model, b::RsysBatch)    # caller wraps this in GC.@preserve of the arrays b points to
    check(ccall((:rsys_batch_upload, LIB), Int32, (Ptr{Cvoid}, Ref{RsysBatch}), m.h, b))
end
# the next batch beside the running step (second staging buffer + copy stream), then made the resident one: enqueue the step,
# prefetch!, read the losses, swap_batch!
function prefetch!(m::Model, b::RsysBatch)  # caller wraps this in GC.@preserve of the arrays b points to
    check(ccall((:rsys_batch_prefetch, LIB), Int32, (Ptr{Cvoid}, Ref{RsysBatch}), m.h, b))
end
swap_batch!(m::Model) = check(ccall((:rsys_batch_swap, LIB), Int32, (Ptr{Cvoid},), m.h))
function forward_backward!(m::Model, evaluate::Bool, task_w::NTuple{4,Float32}, grad_scale::Float32, seed::UInt64, step::UInt64)
    tw = Ref(task_w)
    check(ccall((:rsys_forward_backward, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Float32}, Float32, UInt64, UInt64),
                m.h, evaluate ? 1 : 0, tw, grad_scale, seed, step))
end
function losses(m::Model)
    lo = Vector{Float32}(undef, 12); ws = Vector{Float32}(undef, 4)
    check(ccall((:rsys_losses_get, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), m.h, lo, ws))
    lo, ws
end

# losses without a host wait per step (transformer.py:245-262 accumulates on the device): park, then read every parked step at once
push_losses!(m::Model) = check(ccall((:rsys_losses_push, LIB), Int32, (Ptr{Cvoid},), m.h))
function drain_losses(m::Model, cap::Integer = 1024)
    lo = Matrix{Float32}(undef, 12, cap); ws = Matrix{Float32}(undef, 4, cap); n = Ref{Int32}(0)
    GC.@preserve lo ws check(ccall((:rsys_losses_drain, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Ref{Int32}), m.h, lo, ws, cap, n))
    lo[:, 1:n[]], ws[:, 1:n[]]
end

set_deterministic!(m::Model, on::Bool = true) = check(ccall((:rsys_model_set_deterministic, LIB), Int32, (Ptr{Cvoid}, Int32), m.h, on ? 1 : 0))

# inference forward (model.py:531-538): task 0 = retrieval, 1 = ranking; `tokens` = flat token indices (0-based) to report
function infer_select(m::Model, task::Integer, tokens::Vector{Int32}, D::Integer)
    out = task == 0 ? Matrix{Float32}(undef, D, length(tokens)) : Vector{Float32}(undef, length(tokens))
    GC.@preserve tokens out check(ccall((:rsys_infer_select, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}, Int64, Ptr{Float32}, Int64),
                                        m.h, task, tokens, length(tokens), out, length(out)))
    out
end

function create_optimizer(m::Model; lr = 1f-4, betas = (0.9f0, 0.95f0), eps = 1f-8, weight_decay = 0.1f0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rsys_adamw_create, LIB), Int32, (Ptr{Cvoid}, Float32, Float32, Float32, Float32, Float32, Ref{Ptr{Cvoid}}),
                m.h, lr, betas[1], betas[2], eps, weight_decay, h))
    Optimizer(h[])
end
step!(o::Optimizer; lr_factor = 1f0, clip = 1f0, grad_div = 1f0) =
    check(ccall((:rsys_adamw_step, LIB), Int32, (Ptr{Cvoid}, Float32, Float32, Float32), o.h, lr_factor, clip, grad_div))
# (beyond the reference, opt-in) ZeRO-1: moments for this rank's 1/world of the parameters; step_zero1! replaces allreduce_grads! + step!
set_zero1!(o::Optimizer, rank::Integer, world::Integer) =
    check(ccall((:rsys_adamw_set_zero1, LIB), Int32, (Ptr{Cvoid}, Int32, Int32), o.h, rank, world))
step_zero1!(o::Optimizer, c; lr_factor = 1f0, clip = 1f0, grad_div = 1f0) =
    check(ccall((:rsys_adamw_step_zero1, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float32, Float32, Float32), o.h, c.h, lr_factor, clip, grad_div))

function unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:rsys_comm_unique_id, LIB), Int32, (Ptr{UInt8},), id)); id
end
function Comm(id::Vector{UInt8}, rank::Integer, world::Integer, device::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rsys_comm_init, LIB), Int32, (Ptr{UInt8}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}), id, rank, world, device, h))
    Comm(h[], world)
end
self_test(c::Comm) = check(ccall((:rsys_self_test, LIB), Int32, (Ptr{Cvoid},), c.h))
allreduce_grads!(m::Model, c::Comm) = check(ccall((:rsys_allreduce_grads, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), m.h, c.h))
# arm the early gradient buckets for the next backward (the last micro-step of an optimizer step)
set_split_table_reduce!(m::Model, on::Bool=true) = check(ccall((:rsys_model_set_split_table_reduce, LIB), Int32, (Ptr{Cvoid}, Int32), m.h, on ? 1 : 0))
begin_grad_sync!(m::Model, c::Comm) = check(ccall((:rsys_set_grad_sync, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), m.h, c.h))

# row-sharded item table (rsys_config.table_shard_world >= 1): the communicator of the row exchange and the vocabulary-parallel
# cross entropy, and the table rows [lo, hi) this model holds
set_shard_comm!(m::Model, c::Comm) = check(ccall((:rsys_model_set_shard_comm, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), m.h, c.h))
function table_rows(m::Model)
    lo = Ref{Int64}(0); hi = Ref{Int64}(0)
    check(ccall((:rsys_table_rows, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), m.h, lo, hi)); (lo[], hi[])
end


# ---- lifetime, library
version() = unsafe_string(ccall((:rsys_version, LIB), Cstring, ()))
function device_count()
    n = Ref{Int32}(0); check(ccall((:rsys_device_count, LIB), Int32, (Ref{Int32},), n)); Int(n[])
end
synchronize() = check(ccall((:rsys_device_synchronize, LIB), Int32, ()))
destroy!(m::Model) = (m.h == C_NULL || check(ccall((:rsys_model_destroy, LIB), Int32, (Ptr{Cvoid},), m.h)); m.h = C_NULL; nothing)
destroy!(o::Optimizer) = (o.h == C_NULL || check(ccall((:rsys_adamw_destroy, LIB), Int32, (Ptr{Cvoid},), o.h)); o.h = C_NULL; nothing)
destroy!(c::Comm) = (c.h == C_NULL || check(ccall((:rsys_comm_destroy, LIB), Int32, (Ptr{Cvoid},), c.h)); c.h = C_NULL; nothing)

# ---- parameters (state-dict names of transformer.model.py:346-359)
random_pretrained_embeddings!(m::Model, seed::Integer) = check(ccall((:rsys_model_random_metadata, LIB), Int32, (Ptr{Cvoid}, UInt64), m.h, seed))
function set_rope!(m::Model, c::Matrix{Float32}, s::Matrix{Float32})   # (head_dim / 2, n_pos) column-major = (n_pos, head_dim / 2) row-major
    GC.@preserve c s check(ccall((:rsys_model_set_rope, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int64), m.h, c, s, size(c, 2)))
end
function param_count(m::Model)
    n = Ref{Int32}(0); check(ccall((:rsys_param_count, LIB), Int32, (Ptr{Cvoid}, Ref{Int32}), m.h, n)); Int(n[])
end
function param_info(m::Model, i::Integer)            # i is 0-based like the C ABI
    name = Vector{UInt8}(undef, 256); shape = Vector{Int64}(undef, 2); nd = Ref{Int32}(0); tr = Ref{Int32}(0)
    GC.@preserve name shape check(ccall((:rsys_param_info, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{UInt8}, Csize_t, Ptr{Int64}, Ref{Int32}, Ref{Int32}),
                                        m.h, i, name, length(name), shape, nd, tr))
    (unsafe_string(pointer(name)), nd[] == 1 ? (shape[1],) : (shape[1], shape[2]), tr[] != 0)
end
function grad!(m::Model, name::String, out::Array{Float32})
    GC.@preserve out check(ccall((:rsys_grad_get, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Float32}, Int64), m.h, name, out, length(out)))
    out
end
zero_grad!(m::Model) = check(ccall((:rsys_zero_grad, LIB), Int32, (Ptr{Cvoid},), m.h))
refresh_shadow!(m::Model) = check(ccall((:rsys_refresh_shadow, LIB), Int32, (Ptr{Cvoid},), m.h))
function grad_buffer(m::Model)
    p = Ref{Ptr{Cvoid}}(C_NULL); n = Ref{Int64}(0)
    check(ccall((:rsys_grad_buffer, LIB), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}), m.h, p, n)); (p[], n[])
end
function param_buffer(m::Model)
    p = Ref{Ptr{Cvoid}}(C_NULL); n = Ref{Int64}(0)
    check(ccall((:rsys_param_buffer, LIB), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}), m.h, p, n)); (p[], n[])
end

# ---- clip, heads, serving
function clip_grad_norm!(m::Model, max_norm::Real)    # transformer.py:273
    out = Ref{Float32}(0f0)
    check(ccall((:rsys_clip_grad_norm, LIB), Int32, (Ptr{Cvoid}, Float32, Ref{Float32}), m.h, max_norm, out)); out[]
end
function head_rows(m::Model)
    out = Vector{Int32}(undef, 4)
    GC.@preserve out check(ccall((:rsys_head_rows_get, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}), m.h, out)); out
end
function item_table(m::Model, V::Integer, D::Integer)  # register.py:27-33: (D, V) column-major = (V, D) row-major
    out = Matrix{Float32}(undef, D, V)
    GC.@preserve out check(ccall((:rsys_item_table, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64), m.h, out, length(out))); out
end
function infer(m::Model, task::Integer, rows::Integer, S::Integer, D::Integer)   # model.py:531-538 over every token of the resident batch
    out = task == 0 ? Array{Float32}(undef, D, 2S, rows) : Array{Float32}(undef, 2S, rows)
    GC.@preserve out check(ccall((:rsys_infer, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Float32}, Int64), m.h, task, out, length(out))); out
end
function trunk_output(m::Model, rows::Integer, S::Integer, D::Integer)
    out = Array{Float32}(undef, D, 2S, rows)
    GC.@preserve out check(ccall((:rsys_trunk_output_get, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64), m.h, out, length(out))); out
end

# ---- optimizer state (checkpoint / resume, transformer.py:456-466,690-695)
function adamw_state(o::Optimizer, name::String, n::Integer)
    m = Vector{Float32}(undef, n); v = Vector{Float32}(undef, n); st = Ref{Int32}(0)
    GC.@preserve m v check(ccall((:rsys_adamw_state_get, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Float32}, Ptr{Float32}, Int64, Ref{Int32}),
                                 o.h, name, m, v, n, st))
    (m, v, Int(st[]))
end
function adamw_state!(o::Optimizer, name::String, m::Vector{Float32}, v::Vector{Float32}, step::Integer)
    GC.@preserve m v check(ccall((:rsys_adamw_state_set, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Float32}, Ptr{Float32}, Int64, Int32),
                                 o.h, name, m, v, length(m), step))
end

# ---- collectives beyond the gradient all-reduce
function allreduce_f64!(c::Comm, x::Vector{Float64})   # reduce_mean, transformer.py:199-204
    GC.@preserve x check(ccall((:rsys_allreduce_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32), c.h, x, length(x))); x
end
function grad_sync_early(m::Model)
    n = Ref{Int64}(0); check(ccall((:rsys_grad_sync_early, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}), m.h, n)); n[]
end
# the gradient reduction's bucket schedule of the last optimizer step: rows (first element, one past the last, phase)
function grad_sync_schedule(m::Model; cap::Integer = 64)
    out = zeros(Int64, 3 * cap); n = Ref{Int32}(0)
    GC.@preserve out check(ccall((:rsys_grad_sync_schedule, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int32, Ref{Int32}), m.h, out, cap, n))
    permutedims(reshape(out[1:3 * min(Int(n[]), cap)], 3, :))
end
# (rank, world, transport, RCCL version code)
function comm_info(c::Comm)
    out = zeros(Int32, 4)
    GC.@preserve out check(ccall((:rsys_comm_info, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}), c.h, out)); out
end
# replica consistency (DDP's parameter broadcast, transformer.py:678-682, replaced by same-seed init + comparison): this rank's four
# checksum words travel in one slot per rank of a SUM all-reduce (the other slots zero: exact), then every rank compares all slots
function param_checksum(m::Model)
    out = zeros(Float64, 4)
    GC.@preserve out check(ccall((:rsys_param_checksum, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), m.h, out)); out
end
function assert_replicas_equal(m::Model, c::Comm, rank::Integer, what::String = "")
    w = param_checksum(m); slots = zeros(Float64, 4 * c.world); slots[4rank + 1:4rank + 4] = w
    allreduce_f64!(c, slots)
    all(slots[4r + 1:4r + 4] == w for r in 0:c.world - 1) || error("replicas differ $what: $(reshape(slots, 4, :))")
end

# ---- instrumentation
step_mark!(m::Model) = check(ccall((:rsys_step_mark, LIB), Int32, (Ptr{Cvoid},), m.h))
function step_marks(m::Model, cap::Integer = 65536)
    ms = Vector{Float32}(undef, cap); n = Ref{Int32}(0)
    GC.@preserve ms check(ccall((:rsys_step_marks_get, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int32, Ref{Int32}), m.h, ms, cap, n)); ms[1:n[]]
end
# RSYS_* environment switches (csrc/switches.hpp): parsed at model / communicator creation; reload_switches!() parses on demand
reload_switches!() = check(ccall((:rsys_switches_reload, LIB), Int32, ()))
function switches_set()
    buf = Vector{UInt8}(undef, 4096)
    GC.@preserve buf ccall((:rsys_switches_describe, LIB), Int32, (Ptr{UInt8}, Int32), buf, length(buf))
    unsafe_string(pointer(buf))
end
op_timing!(m::Model, mode::Integer) = check(ccall((:rsys_op_timing, LIB), Int32, (Ptr{Cvoid}, Int32), m.h, mode))
op_timing_filter!(m::Model, substr::AbstractString) = check(ccall((:rsys_op_timing_filter, LIB), Int32, (Ptr{Cvoid}, Cstring), m.h, substr))
function timing_report(m::Model)
    buf = Vector{UInt8}(undef, 1 << 16)
    GC.@preserve buf check(ccall((:rsys_timing_get, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Csize_t), m.h, buf, length(buf)))
    unsafe_string(pointer(buf))
end

# One optimizer step of train_epoch (transformer.py:256-276) with grad_accum = 1
function train_step!(m::Model, o::Optimizer, c::Union{Comm,Nothing}, task_w, lr_factor, seed, step)
    c === nothing || begin_grad_sync!(m, c)
    forward_backward!(m, false, task_w, 1f0, seed, step)
    c === nothing || allreduce_grads!(m, c)
    step!(o; lr_factor = Float32(lr_factor), clip = 1f0, grad_div = Float32(c === nothing ? 1 : c.world))
end

end # module
