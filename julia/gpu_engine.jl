# gpu_engine.jl -- DPMMSubClusters.jl on MI355X, master AND worker side native (the recommended binding).
#
# `include` this file from src/DPMMSubClusters.jl (after ds.jl / the priors: it uses niw_hyperparams, multinomial_hyper and the schedule
# constants of global_params.jl).  It adds `gpu_fit` / `gpu_dp_parallel` next to the reference's `fit` / `dp_parallel`
# (src/dp-parallel-sampling.jl:121-157, 215-293) with the same positional arguments, keyword names, defaults and return tuples; one
# Julia process per GPU.  Everything inside a sweep is ONE ccall: dpmmh_group_step (include/dpmm_host.h), which drives libdpmmhip.so
# (include/dpmm_hip.h, dpmm_hip_master.h) through a table of C function pointers.
#
# UNEXECUTED: the build image has no Julia.  Written against DPMMH_ABI_VERSION 5 / DPMM_ABI_VERSION 3; the same call sequence runs end to
# end through the ctypes binding (dpmmsubclusters.jl_amd/host/engine.py), and tests/test_integration_layout.py checks this file
# mechanically: every symbol it names is exported, the WorkerTable has the members of struct dpmmh_worker in header order, every ccall
# passes as many arguments as its C prototype declares.
using LinearAlgebra
using Libdl
const libhip  = Libdl.dlopen("libdpmmhip.so")
const libhost = Libdl.dlopen("libdpmmhost.so")
hip(sym) = Libdl.dlsym(libhip, sym)

# struct dpmmh_worker: ctx, rank, world, then 13 + 11 + 4 + 2 function pointers in the order of include/dpmm_host.h (the eleven niw_* /
# *_device entries -- the NIW master's dense maths on the device -- and the eight mult_* entries -- the Multinomial master's Dirichlet
# draws and log-marginals on the device -- are optional, each as a group: C_NULL for all of a group keeps that work on the host)
struct WorkerTable
    ctx::Ptr{Cvoid}; rank::Cint; world::Cint
    params_staging::Ptr{Cvoid}; commit_params::Ptr{Cvoid}; set_num_clusters::Ptr{Cvoid}; sweep::Ptr{Cvoid}
    step_stats::Ptr{Cvoid}; stats::Ptr{Cvoid}; split::Ptr{Cvoid}; merge::Ptr{Cvoid}; remove_empty::Ptr{Cvoid}
    reset_sublabels::Ptr{Cvoid}; init_labels::Ptr{Cvoid}; allgather::Ptr{Cvoid}; last_error::Ptr{Cvoid}
    niw_master_setup::Ptr{Cvoid}; step_stats_device::Ptr{Cvoid}; step_master_device::Ptr{Cvoid}; stats_device::Ptr{Cvoid}; niw_posterior::Ptr{Cvoid}
    niw_draw::Ptr{Cvoid}; niw_pairs::Ptr{Cvoid}; niw_pairs_ahead::Ptr{Cvoid}; niw_put_rows::Ptr{Cvoid}; niw_rows::Ptr{Cvoid}; niw_draws::Ptr{Cvoid}
    mult_master_setup::Ptr{Cvoid}; mult_draw::Ptr{Cvoid}; mult_draws::Ptr{Cvoid}; mult_put_rows::Ptr{Cvoid}
    mult_pairs_ahead::Ptr{Cvoid}; mult_marginals::Ptr{Cvoid}; mult_rows_on_demand::Ptr{Cvoid}; mult_rows_wait::Ptr{Cvoid}
end
native_table(ctx, rank, world) = WorkerTable(ctx, rank, world,
    hip(:dpmm_params_staging), hip(:dpmm_commit_params), hip(:dpmm_set_num_clusters), hip(:dpmm_sweep),
    hip(:dpmm_step_stats), hip(:dpmm_suffstats_host), hip(:dpmm_split), hip(:dpmm_merge), hip(:dpmm_remove_empty),
    hip(:dpmm_reset_sublabels), hip(:dpmm_init_labels_from), hip(:dpmm_comm_allgather_host), hip(:dpmm_last_error),
    hip(:dpmm_niw_master_setup), hip(:dpmm_step_stats_device), hip(:dpmm_step_master_device), hip(:dpmm_suffstats_device), hip(:dpmm_niw_master_posterior),
    hip(:dpmm_niw_master_draw), hip(:dpmm_niw_master_pairs), hip(:dpmm_niw_master_pairs_ahead), hip(:dpmm_niw_master_put_rows), hip(:dpmm_niw_master_rows), hip(:dpmm_niw_master_draws),
    hip(:dpmm_mult_master_setup), hip(:dpmm_mult_master_draw), hip(:dpmm_mult_master_draws), hip(:dpmm_mult_master_put_rows),
    hip(:dpmm_mult_master_pairs_ahead), hip(:dpmm_mult_master_marginals), hip(:dpmm_mult_master_rows_on_demand), hip(:dpmm_mult_master_rows_wait))

# fit(all_data::AbstractArray{Float32,2}, hyper::niw_hyperparams, α; iters, init_clusters, seed, burnout, ...)  -- one process per GPU
function gpu_fit(pts::Matrix{Float32}, hyper::niw_hyperparams, α::Float32; iters = 100, init_clusters = 1, seed = 1, burnout = 20,
                 first_index = 0, total = size(pts, 2), device = 0, rank = 0, world = 1, uid = nothing)
    D, n = size(pts)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    @assert 0 == ccall(hip(:dpmm_create), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Int64, Int64, Cint, UInt64), ctx, 0, D, n, first_index, device, seed)
    @assert 0 == ccall(hip(:dpmm_upload_points), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), ctx[], pts, D)     # D x n column-major == the ABI layout
    world > 1 && @assert 0 == ccall(hip(:dpmm_comm_init), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), ctx[], uid, rank, world)   # uid: 128 bytes from rank 0's dpmm_comm_unique_id
    model = Ref{Ptr{Cvoid}}(C_NULL)
    @assert 0 == ccall(Libdl.dlsym(libhost, :dpmmh_model_create), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cdouble, Int64, UInt64, Cint, Cint),
                       model, 0, D, α, total, seed, burnout, Sys.CPU_THREADS ÷ max(world, 1))
    @assert 0 == ccall(Libdl.dlsym(libhost, :dpmmh_model_set_prior_niw), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ptr{Float64}, Cdouble, Ptr{Float64}),
                       model[], 0, hyper.κ, hyper.m, hyper.ν, hyper.ψ)          # ψ symmetric: column- and row-major agree
    table = Ref(native_table(ctx[], rank, world))
    @assert 0 == ccall(Libdl.dlsym(libhost, :dpmmh_model_bind_worker), Cint, (Ptr{Cvoid}, Ref{WorkerTable}), model[], table)
    @assert 0 == ccall(Libdl.dlsym(libhost, :dpmmh_model_init_first_clusters), Cint, (Ptr{Cvoid}, Cint), model[], init_clusters)
    step = Libdl.dlsym(libhost, :dpmmh_group_step)
    iter_count = Float64[]
    for i in 1:iters                                   # run_model, src/dp-parallel-sampling.jl:351-404
        final = i >= iters - argmax_sample_stop
        no_more_splits = i >= iters - split_stop      # (|| length(clusters) >= max_clusters: query "K" with dpmmh_model_get)
        push!(iter_count, @elapsed @assert 0 == ccall(step, Cint, (Ptr{Cvoid}, Cint, Cint), model[], no_more_splits, final))
    end
    labels = Vector{Int64}(undef, n); sub = Vector{Int64}(undef, n)
    ccall(hip(:dpmm_get_labels), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), ctx[], labels, sub)
    return labels, sub, iter_count, model[], ctx[]     # cluster state: dpmmh_model_get(model, "weights" | "mu" | "R" | "N" | ...)
end


# ---- the reference's entry points over the native engine -------------------------------------------------------------------------
host(sym) = Libdl.dlsym(libhost, sym)

function model_get(model::Ptr{Cvoid}, field::String, ::Type{T}, dims...) where {T}
    out = Array{T}(undef, dims...)
    nb = ccall(host(:dpmmh_model_get), Int64, (Ptr{Cvoid}, Cstring, Ptr{Cvoid}, Int64), model, field, out, sizeof(out))
    nb == sizeof(out) || error("dpmmh_model_get($field): $nb bytes, expected $(sizeof(out))")
    return out
end

# dp_parallel(all_data, local_hyper_params, α_param, iters, init_clusters, seed, verbose, save_model, burnout, gt, max_clusters, ...)
# src/dp-parallel-sampling.jl:121-157 -- returns (dp_model, iter_count, nmi_score_history, liklihood_history, cluster_count_history)
function gpu_dp_parallel(all_data::AbstractArray{Float32,2}, local_hyper_params::niw_hyperparams, α_param::Float32,
                         iters::Int64 = 100, init_clusters::Int64 = 1, seed = nothing, verbose = true, save_model = false,
                         burnout = 15, gt = nothing, max_clusters = Inf; device = 0)
    sd = seed === nothing ? rand(UInt64) : UInt64(seed)
    labels, sub, iter_count, model, ctx = gpu_fit(Matrix{Float32}(all_data), local_hyper_params, α_param; iters = iters,
                                                  init_clusters = init_clusters, seed = sd, burnout = burnout, device = device)
    K = Int(model_get(model, "K", Int64, 1)[1])
    weights = model_get(model, "weights", Float32, K)
    return (labels = labels, labels_subcluster = sub, weights = weights, model = model, ctx = ctx), iter_count, Float64[], Float64[], Int[]
end

# fit(all_data, local_hyper_params, α_param; iters, init_clusters, seed, verbose, save_model, burnout, gt, max_clusters, ...)
# src/dp-parallel-sampling.jl:215-219 -- the 9-tuple (labels, clusters, weights, iter_count, nmi, likelihood, cluster_count, sub_labels)
function gpu_fit_reference_shape(all_data::AbstractArray{Float32,2}, local_hyper_params::niw_hyperparams, α_param::Float32;
                                 iters = 100, init_clusters = 1, seed = nothing, verbose = true, save_model = false, burnout = 20,
                                 gt = nothing, max_clusters = Inf, device = 0)
    dp_model, iter_count, nmi, lik, kh = gpu_dp_parallel(all_data, local_hyper_params, α_param, iters, init_clusters, seed, verbose,
                                                         save_model, burnout, gt, max_clusters; device = device)
    D = size(all_data, 1); K = length(dp_model.weights)
    mu = model_get(dp_model.model, "mu", Float32, D, 3K)                 # [3K][D] row-major == (D, 3K) column-major
    R  = model_get(dp_model.model, "R", Float32, D, D, 3K)               # [3K][D][D] row-major: R[:, :, j]' is the upper-triangular factor
    clusters = [(μ = mu[:, 3k - 2], R = permutedims(R[:, :, 3k - 2])) for k in 1:K]     # cluster-level rows 3(k-1) of the ABI (0-based)
    return dp_model.labels, clusters, dp_model.weights, iter_count, nmi, lik, kh, dp_model.labels_subcluster
end
# default prior of fit(all_data, α_param; ...), src/dp-parallel-sampling.jl:270-274
gpu_fit_reference_shape(all_data::AbstractArray{Float32,2}, α_param::Float32; kw...) =
    gpu_fit_reference_shape(all_data, niw_hyperparams(1.0f0, zeros(Float64, size(all_data, 1)), Float32(size(all_data, 1) + 3),
                                                      Matrix{Float64}(I, size(all_data, 1), size(all_data, 1))), α_param; kw...)
