# gpu_worker.jl -- DPMMSubClusters.jl on MI355X, worker side only: Julia's own master code (sample_clusters!, check_and_split!,
# check_and_merge!, ...) stays as it is and the @spawnat ...worker!(...) calls of src/local_clusters_actions.jl become calls on the
# GpuWorker each Julia worker process owns.  `include` from src/DPMMSubClusters.jl; uses the types of ds.jl and the priors.
#
# This is the DROP-IN surface of include/dpmm_hip.h alone (no dpmm_hip_master.h, no debug entry): parameters in the reference's own
# thin_cluster_params form, statistics back in the reference's thin_suff_stats form.
#
# UNEXECUTED: the build image has no Julia.  Written against DPMM_ABI_VERSION 3; tests/test_integration_layout.py parses this file: the
# array comprehensions of set_params! are emulated and compared with the ABI's memory layouts, every symbol is checked against the
# header, every ccall's argument count against its C prototype.
const libdpmm = "libdpmmhip.so"          # on LD_LIBRARY_PATH / dlopen path

mutable struct GpuWorker
    h::Ptr{Cvoid}; D::Int; n::Int; K::Int; stride::Int
end

function dpmm_check(rc::Cint, w)
    rc == 0 && return
    msg = unsafe_string(ccall((:dpmm_last_error, libdpmm), Cstring, (Ptr{Cvoid},), w === nothing ? C_NULL : w.h))
    error("libdpmmhip: $msg (code $rc)")
end

prior_kind(::niw_hyperparams) = Cint(0)
prior_kind(::multinomial_hyper) = Cint(1)

function GpuWorker(hyper::distribution_hyper_params, pts::AbstractArray{Float32,2}, first_index::Int, device::Int, seed)
    D, n = size(pts)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:dpmm_create, libdpmm), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Int64, Int64, Cint, UInt64),
               ref, prior_kind(hyper), D, n, first_index, device, UInt64(seed))
    dpmm_check(rc, nothing)
    w = GpuWorker(ref[], D, n, 0, 0)
    finalizer(x -> ccall((:dpmm_destroy, libdpmm), Cint, (Ptr{Cvoid},), x.h), w)
    # points are D x n column-major Float32 == the ABI's layout with ldx = D
    dpmm_check(ccall((:dpmm_upload_points, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), w.h, pts, D), w)
    w.stride = ccall((:dpmm_packed_stride, libdpmm), Int64, (Ptr{Cvoid},), w.h)
    return w
end

# replaces broadcast_cluster_params(params_vector, weights) for mv_gaussian clusters
function set_params!(w::GpuWorker, params::Vector{thin_cluster_params{mv_gaussian}}, weights::Vector{Float32})
    K = length(params); D = w.D
    dists(p) = (p.cluster_dist, p.l_dist, p.r_dist)
    # flattened generators nest left to right (rightmost `for` runs fastest): memory order [p][d][i] == the ABI's [3K][D]
    mu  = Float32[d.μ[i]        for p in params for d in dists(p) for i in 1:D]
    inv = Float32[d.invΣ[i, j]  for p in params for d in dists(p) for j in 1:D for i in 1:D]   # [3K][D][D]
    ld  = Float32[d.logdetΣ     for p in params for d in dists(p)]
    lr  = Float32[p.lr_weights[s] for s in 1:2, p in params]
    dpmm_check(ccall((:dpmm_set_params_niw, libdpmm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
        w.h, K, mu, inv, ld, lr, weights), w)
    w.K = K
end

function set_params!(w::GpuWorker, params::Vector{thin_cluster_params{multinomial_dist}}, weights::Vector{Float32})
    K = length(params)
    logp = Float32[d.α[i] for p in params for d in (p.cluster_dist, p.l_dist, p.r_dist) for i in 1:w.D]   # [3K][D]
    lr   = Float32[p.lr_weights[s] for s in 1:2, p in params]
    dpmm_check(ccall((:dpmm_set_params_mult, libdpmm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}), w.h, K, logp, lr, weights), w)
    w.K = K
end

# replaces sample_labels!(group, final, …) followed by sample_sub_clusters!(group)
sweep!(w::GpuWorker, epoch::Integer, final::Bool) =
    dpmm_check(ccall((:dpmm_sweep, libdpmm), Cint, (Ptr{Cvoid}, UInt32, Cint), w.h, epoch, final), w)

# replaces create_suff_stats_dict_worker(…, indices): returns Dict(index => thin_suff_stats)
function suff_stats(w::GpuWorker, hyper, indices)
    packed = Matrix{Float64}(undef, w.stride, 2 * w.K)
    idx = indices === nothing ? C_NULL : Int64.(indices)
    dpmm_check(ccall((:dpmm_suffstats_packed, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Int64}, Cint, Ptr{Float64}),
                     w.h, idx, indices === nothing ? 0 : length(indices), packed), w)
    N = Array{Float64}(undef, 3, w.K); s = Array{Float64}(undef, w.D, 3, w.K); S = Array{Float64}(undef, w.D, w.D, 3, w.K)
    dpmm_check(ccall((:dpmm_unpack_suffstats, libdpmm), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        w.h, w.K, packed, N, s, hyper isa niw_hyperparams ? S : C_NULL), w)
    mk(k, c) = hyper isa niw_hyperparams ?
        niw_sufficient_statistics(N[c, k], s[:, c, k], S[:, :, c, k]) :
        multinomial_sufficient_statistics(N[c, k], Float32.(s[:, c, k]))
    ks = indices === nothing ? (1:w.K) : indices
    return Dict(k => thin_suff_stats(mk(k, 1), mk(k, 2), mk(k, 3)) for k in ks)
end

split!(w::GpuWorker, idx::Vector{Int64}, new_idx::Vector{Int64}, epoch) =
    dpmm_check(ccall((:dpmm_split, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint, UInt32), w.h, idx, new_idx, length(idx), epoch), w)
merge!(w::GpuWorker, idx::Vector{Int64}, new_idx::Vector{Int64}) =
    dpmm_check(ccall((:dpmm_merge, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), w.h, idx, new_idx, length(idx)), w)
remove_empty!(w::GpuWorker, pts_count::Vector{Int64}) =
    dpmm_check(ccall((:dpmm_remove_empty, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Int64}, Cint), w.h, pts_count, length(pts_count)), w)
reset_sublabels!(w::GpuWorker, idx, epoch) =
    dpmm_check(ccall((:dpmm_reset_sublabels, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Int64}, Cint, UInt32),
                     w.h, idx === nothing ? C_NULL : idx, idx === nothing ? 0 : length(idx), epoch), w)
function labels(w::GpuWorker)
    l = Vector{Int64}(undef, w.n); s = Vector{Int64}(undef, w.n)
    dpmm_check(ccall((:dpmm_get_labels, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), w.h, l, s), w)
    return l, s
end
function bin_counts(w::GpuWorker)        # (2,K): N of the left / right statistics, before the statistics pass
    c = Matrix{Int64}(undef, 2, w.K)
    dpmm_check(ccall((:dpmm_bin_counts, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Int64}), w.h, c), w)
    return c
end
# smart splits: the worker calls inside smart_cluster_init! (v, mu::Vector{Float64})
function smart_project(w::GpuWorker, k, v, mu)
    vals = Vector{Float64}(undef, w.n); cnt = Ref{Int64}(0)
    dpmm_check(ccall((:dpmm_smart_project, libdpmm), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}),
                     w.h, k, v, mu, vals, cnt), w)
    return resize!(vals, cnt[])          # percentile(vals, 0.10), percentile(vals, 0.90) as in :650
end
function kmeans_iter(w::GpuWorker, k, lo, hi)
    out = zeros(Float64, 4)              # (sum, count) of the lo side, (sum, count) of the hi side
    dpmm_check(ccall((:dpmm_smart_kmeans_iter, libdpmm), Cint, (Ptr{Cvoid}, Int64, Float64, Float64, Ptr{Float64}), w.h, k, lo, hi, out), w)
    return (out[1], out[2]), (out[3], out[4])
end
smart_assign!(w::GpuWorker, k, lo, hi) =
    dpmm_check(ccall((:dpmm_smart_assign, libdpmm), Cint, (Ptr{Cvoid}, Int64, Float64, Float64), w.h, k, lo, hi), w)
# advanced mode: rows of the Samples x Dimensions .npy (npzread keeps it column-major, so pass permutedims or an mmap
# of the file body); Float64 files need no Float32.(...) copy on the host
upload_npy!(w::GpuWorker, rows::Matrix{Float64}, ld) =
    dpmm_check(ccall((:dpmm_upload_points_npy, libdpmm), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Cint), w.h, rows, 1, ld, 1), w)


# ---- where the master changes (src/local_clusters_actions.jl:658-673) --------------------------------------------------------------
# group_step with workers::Vector of GpuWorker (one per Julia worker process, `device = myid() - 2`), everything else unchanged:
#
#   sample_clusters!(group, false)
#   foreach(w -> set_params!(w, [create_thin_cluster_params(x) for x in group.local_clusters], group.weights), workers)
#   foreach(w -> sweep!(w, next_epoch!(), hard_clustering ? true : final), workers)
#   update_suff_stats_posterior!(group)      # sum of suff_stats(w, hyper, nothing) over the workers -- or, with dpmm_comm_init, one
#                                            # RCCL all-reduce inside the library (step_stats below)
#
# `epoch` is any counter that is unique per randomised call (the library's RNG is counter-based: Philox keyed by `seed`, counter =
# global point index, epoch).

# steps 5 + 6 of group_step in one device pass (dpmm_step_stats): rows summed over the ranks + the bad-cluster flags
function step_stats(w::GpuWorker, reset_epoch)
    packed = Ref{Ptr{Float64}}(C_NULL); bad = Ref{Ptr{UInt8}}(C_NULL)
    dpmm_check(ccall((:dpmm_step_stats, libdpmm), Cint, (Ptr{Cvoid}, UInt32, Ref{Ptr{Float64}}, Ref{Ptr{UInt8}}), w.h, reset_epoch, packed, bad), w)
    return unsafe_wrap(Array, packed[], (w.stride, 2 * w.K)), unsafe_wrap(Array, bad[], w.K)     # valid until the worker's next call
end

# multi-GPU: rank 0 makes the id, every rank joins; afterwards the statistics calls return rows summed over the ranks
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = ccall((:dpmm_comm_unique_id, libdpmm), Cint, (Ptr{UInt8},), id); rc == 0 || error("dpmm_comm_unique_id: $rc")
    return id
end
comm_init!(w::GpuWorker, id::Vector{UInt8}, rank, world) =
    dpmm_check(ccall((:dpmm_comm_init, libdpmm), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), w.h, id, rank, world), w)
