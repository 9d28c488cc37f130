# gpu_engine.jl -- DPMMSubClusters.jl on MI355X, master AND worker side native (the recommended binding).
#
# `include` this file from src/DPMMSubClusters.jl (after ds.jl / the priors: it uses niw_hyperparams, multinomial_hyper and the schedule
# constants of global_params.jl).  It adds `gpu_fit` / `gpu_dp_parallel` next to the reference's `fit` / `dp_parallel`
# (src/dp-parallel-sampling.jl:121-157, 215-293) with the same positional arguments, keyword names, defaults and return tuples -- for
# every `distribution_hyper_params` the engine implements (niw_hyperparams, multinomial_hyper) -- and `run_model!` with the reference's
# four histories (iteration times, NMI from an on-device contingency table when `gt` is given, log posterior when `verbose`, cluster
# counts), `max_clusters`, the outlier component, smart splits and `save_model` checkpoints; one Julia process per GPU.  Everything inside a sweep is ONE ccall: dpmmh_group_step (include/dpmm_host.h), which drives libdpmmhip.so
# (include/dpmm_hip.h, dpmm_hip_master.h) through a table of C function pointers.
#
# UNEXECUTED: the build image has no Julia.  Written against DPMMH_ABI_VERSION 5 / DPMM_ABI_VERSION 3; the same call sequence runs end to
# end through the ctypes binding (dpmmsubclusters.jl_amd/host/engine.py), and tests/test_integration_layout.py checks this file
# mechanically: every symbol it names is exported, the WorkerTable has the members of struct dpmmh_worker in header order, every ccall
# passes as many arguments as its C prototype declares.
using LinearAlgebra
using Libdl
using Serialization
const libhip  = Libdl.dlopen("libdpmmhip.so")
const libhost = Libdl.dlopen("libdpmmhost.so")
hip(sym) = Libdl.dlsym(libhip, sym)
@assert ccall(hip(:dpmm_abi_version), Cint, ()) == 3 "libdpmmhip.so: DPMM_ABI_VERSION differs from the one this file is written for (include/dpmm_hip.h)"
@assert ccall(Libdl.dlsym(libhost, :dpmmh_abi_version), Cint, ()) == 5 "libdpmmhost.so: DPMMH_ABI_VERSION differs from the one this file is written for (include/dpmm_host.h)"

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

host(sym) = Libdl.dlsym(libhost, sym)
chk(rc, what) = rc == 0 || error("$what failed with status $rc")

function model_get(model::Ptr{Cvoid}, field::String, ::Type{T}, dims...) where {T}
    out = Array{T}(undef, dims...)
    nb = ccall(host(:dpmmh_model_get), Int64, (Ptr{Cvoid}, Cstring, Ptr{Cvoid}, Int64), model, field, out, sizeof(out))
    nb == sizeof(out) || error("dpmmh_model_get($field): $nb bytes, expected $(sizeof(out))")
    return out
end
model_K(model) = Int(model_get(model, "K", Int64, 1)[1])

# The engine of one process = one GPU shard: the worker context (libdpmmhip.so) and the native master (libdpmmhost.so) bound to it.
# Keeps the WorkerTable and the split hook alive for as long as the model uses them.
mutable struct GpuEngine
    ctx::Ptr{Cvoid}; model::Ptr{Cvoid}; table::Base.RefValue{WorkerTable}; hook::Any
    D::Int; n::Int; first_index::Int; kind::Int
end

# distribution_hyper_params -> dpmmh_model_set_prior_* (priors/niw.jl:6-11, priors/multinomial_prior.jl:6-8); which = 0 cluster prior, 1 outlier prior
prior_kind(::niw_hyperparams) = 0
prior_kind(::multinomial_hyper) = 1
set_prior!(model, which, h::niw_hyperparams) =
    chk(ccall(host(:dpmmh_model_set_prior_niw), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ptr{Float64}, Cdouble, Ptr{Float64}),
              model, which, Float64(h.κ), Vector{Float64}(h.m), Float64(h.ν), Matrix{Float64}(h.ψ)), "dpmmh_model_set_prior_niw")   # ψ symmetric: column- and row-major agree
set_prior!(model, which, h::multinomial_hyper) =
    chk(ccall(host(:dpmmh_model_set_prior_mult), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}), model, which, Vector{Float32}(h.α)), "dpmmh_model_set_prior_mult")

# smart_cluster_init!(group, cluster_num) (src/local_clusters_actions.jl:555-627) for the clusters the engine names after a split: the master
# half in Julia (direction from the cluster's statistics, the 1-D 2-means loop), the worker halves on the GPU (dpmm_smart_*).  One rank.
function smart_cluster_init!(e::GpuEngine, k::Int, max_split_iter::Int)
    K = model_K(e.model); D = e.D
    N = model_get(e.model, "N", Float64, 3, K)[1, k]
    N > 0 || return
    S = reshape(model_get(e.model, "S", Float64, D, D, 3, K)[:, :, 1, k], D, D)
    μ = model_get(e.model, "sums", Float64, D, 3, K)[:, 1, k] ./ N
    F = eigen(Symmetric(S ./ N .- μ * μ'))
    v = Vector{Float64}(F.vectors[argmax(F.values), :])                 # (a ROW, as the reference takes it: local_clusters_actions.jl:566)
    vals = Vector{Float64}(undef, max(e.n, 1)); cnt = Ref{Int64}(0)
    chk(ccall(hip(:dpmm_smart_project), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}), e.ctx, k, v, μ, vals, cnt), "dpmm_smart_project")
    cnt[] > 1 || return
    t = sort!(vals[1:cnt[]])
    q(p) = (h = (length(t) - 1) * p + 1; lo = floor(Int, h); lo >= length(t) ? t[end] : t[lo] + (h - lo) * (t[lo + 1] - t[lo]))
    lo, hi = q(0.10 * 0.01), q(0.90 * 0.01)                             # percentile(t, 0.10) / (t, 0.90) on StatsBase's 0..100 scale (:574-575)
    it = 0
    while it < max_split_iter
        out = zeros(Float64, 4)
        chk(ccall(hip(:dpmm_smart_kmeans_iter), Cint, (Ptr{Cvoid}, Int64, Cdouble, Cdouble, Ptr{Float64}), e.ctx, k, lo, hi, out), "dpmm_smart_kmeans_iter")
        nlo, nhi = out[1] / out[2], out[3] / out[4]
        (nlo == lo && nhi == hi) && break
        lo, hi = nlo, nhi; it += 1
    end
    chk(ccall(hip(:dpmm_smart_assign), Cint, (Ptr{Cvoid}, Int64, Cdouble, Cdouble), e.ctx, k, lo, hi), "dpmm_smart_assign")
end

# init_model_from_data (src/dp-parallel-sampling.jl:36-53) + init_first_clusters! (:62-78) for this process's shard
function engine_create(pts::Matrix{Float32}, hyper::distribution_hyper_params, α::Float32; init_clusters = 1, seed = 1, burnout = 20,
                       outlier_weight = 0, outlier_params = nothing, smart_splits = false, hard_clustering = false,
                       first_index = 0, total = size(pts, 2), device = 0, rank = 0, world = 1, uid = nothing)
    D, n = size(pts)
    kind = prior_kind(hyper)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ccall(hip(:dpmm_create), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Int64, Int64, Cint, UInt64), ctx, kind, D, n, first_index, device, seed), "dpmm_create")
    chk(ccall(hip(:dpmm_upload_points), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), ctx[], pts, D), "dpmm_upload_points")     # D x n column-major == the ABI layout
    world > 1 && chk(ccall(hip(:dpmm_comm_init), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), ctx[], uid, rank, world), "dpmm_comm_init")   # uid: 128 bytes from rank 0's dpmm_comm_unique_id
    model = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ccall(host(:dpmmh_model_create), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cdouble, Int64, UInt64, Cint, Cint),
              model, kind, D, α, total, seed, burnout, Sys.CPU_THREADS ÷ max(world, 1)), "dpmmh_model_create")
    set_prior!(model[], 0, hyper)
    if outlier_weight > 0                                      # outlier_mod / outlier_hyper_params (global_params.jl, dp-parallel-sampling.jl:63-65)
        set_prior!(model[], 1, outlier_params)
        chk(ccall(host(:dpmmh_model_set_outlier), Cint, (Ptr{Cvoid}, Cdouble), model[], Float64(outlier_weight)), "dpmmh_model_set_outlier")
    end
    hard_clustering && chk(ccall(host(:dpmmh_model_set_option), Cint, (Ptr{Cvoid}, Cint, Cdouble), model[], 1, 1.0), "dpmmh_model_set_option")   # DPMMH_OPT_HARD_CLUSTERING
    table = Ref(native_table(ctx[], rank, world))
    chk(ccall(host(:dpmmh_model_bind_worker), Cint, (Ptr{Cvoid}, Ref{WorkerTable}), model[], table), "dpmmh_model_bind_worker")
    e = GpuEngine(ctx[], model[], table, nothing, D, n, first_index, kind)
    if smart_splits && kind == 0                              # use_smart_splits: Gaussian prior only (local_clusters_actions.jl:555)
        hook = function (user::Ptr{Cvoid}, clusters::Ptr{Int64}, nc::Cint)::Cint
            for j in 1:nc
                smart_cluster_init!(e, Int(unsafe_load(clusters, j)), max_split_iter)
            end
            return Cint(0)
        end
        e.hook = @cfunction($hook, Cint, (Ptr{Cvoid}, Ptr{Int64}, Cint))
        chk(ccall(host(:dpmmh_model_set_split_hook), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), e.model, e.hook, C_NULL), "dpmmh_model_set_split_hook")
    end
    chk(ccall(host(:dpmmh_model_init_first_clusters), Cint, (Ptr{Cvoid}, Cint), e.model, init_clusters), "dpmmh_model_init_first_clusters")
    return e
end

function engine_labels(e::GpuEngine)
    labels = Vector{Int64}(undef, e.n); sub = Vector{Int64}(undef, e.n)
    chk(ccall(hip(:dpmm_get_labels), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), e.ctx, labels, sub), "dpmm_get_labels")
    return labels, sub
end

# mutualinfo(a, b, normed = true) = 2 I / (H_a + H_b) and varinfo = H_a + H_b - 2 I (Clustering.jl, as run_model logs them:
# src/dp-parallel-sampling.jl:370-377) from the K x n_gt contingency table the GPU returns -- instead of gathering N labels per iteration
function nmi_vi(C::AbstractMatrix{Int64})
    N = sum(C); N == 0 && return 0.0, 0.0
    P = C ./ N; pa = sum(P, dims = 2); pb = sum(P, dims = 1)
    I = sum(P[i, j] * log(P[i, j] / (pa[i] * pb[j])) for i in axes(P, 1), j in axes(P, 2) if P[i, j] > 0; init = 0.0)
    H(p) = -sum(x * log(x) for x in p if x > 0; init = 0.0)
    Ha, Hb = H(pa), H(pb)
    return (Ha + Hb > 0 ? 2I / (Ha + Hb) : 1.0), Ha + Hb - 2I
end

# save_model (src/dp-parallel-sampling.jl:428-455) through the engine's state access: everything dpmmh_model_set needs to continue the SAME
# chain (include/dpmm_host.h: K first, then "packed", "lr_weights", "weights", "splittable", "hist", "points_count", "counters") + the labels
function save_checkpoint(e::GpuEngine, path::String, prefix::String, iter::Int, elapsed, burnout::Int)
    K = model_K(e.model)
    stride = e.kind == 0 ? 1 + e.D + (e.D * (e.D + 1)) ÷ 2 : 1 + e.D
    labels, sub = engine_labels(e)
    state = Dict("K" => K, "iter" => iter, "elapsed" => elapsed, "labels" => labels, "labels_subcluster" => sub,
                 "packed" => model_get(e.model, "packed", Float64, stride, 2K), "lr_weights" => model_get(e.model, "lr_weights", Float32, 2, K),
                 "weights" => model_get(e.model, "weights", Float32, K), "splittable" => model_get(e.model, "splittable", UInt8, K),
                 "hist" => model_get(e.model, "hist", Float32, burnout + 5, K), "points_count" => model_get(e.model, "points_count", Int64, K),
                 "counters" => model_get(e.model, "counters", Int64, 8))
    open(io -> serialize(io, state), joinpath(path, prefix * string(iter) * ".jls"), "w")
end

# run_model (src/dp-parallel-sampling.jl:336-404): the iteration loop with the schedule flags, the four histories and the checkpoint rule.
# Returns (iter_count, nmi_score_history, liklihood_history, cluster_count_history); e.vi_history is not kept (the reference returns none).
function run_model!(e::GpuEngine, iters::Int, first_iter::Int = 1; verbose = true, gt = nothing, max_clusters = Inf, save_model = false,
                    burnout = 20, save_path = ".", save_prefix = save_file_prefix, save_interval = model_save_interval)
    step = host(:dpmmh_group_step)
    iter_count = Float64[]; nmi_score_history = Any[]; liklihood_history = Any[]; cluster_count_history = Int[]
    n_gt = 0
    if gt !== nothing                                          # the shard's ground truth goes to the GPU once (ids 0 .. n_gt-1)
        ids = sort!(unique(gt)); code = Dict(v => Int64(i - 1) for (i, v) in enumerate(ids)); n_gt = length(ids)
        mine = Int64[code[g] for g in gt[e.first_index + 1 : e.first_index + e.n]]
        chk(ccall(hip(:dpmm_set_ground_truth), Cint, (Ptr{Cvoid}, Ptr{Int64}, Cint), e.ctx, mine, n_gt), "dpmm_set_ground_truth")
    end
    start_time = time()
    K = model_K(e.model)
    for i in first_iter:iters
        final = i >= iters - argmax_sample_stop                # global_params.jl:10-11
        no_more_splits = i >= iters - split_stop || K >= max_clusters
        dt = @elapsed chk(ccall(step, Cint, (Ptr{Cvoid}, Cint, Cint), e.model, no_more_splits, final), "dpmmh_group_step")
        push!(iter_count, dt)
        K = model_K(e.model)
        push!(cluster_count_history, K)
        vi = "no gt"
        if gt !== nothing
            C = zeros(Int64, n_gt, K)                          # counts[k][g] row-major == (n_gt, K) column-major
            chk(ccall(hip(:dpmm_contingency), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), e.ctx, K, C), "dpmm_contingency")
            nmi, vi = nmi_vi(permutedims(C))                   # (world > 1: sum C over the ranks first -- host_transport_mpi.jl)
            push!(nmi_score_history, nmi)
        else
            push!(nmi_score_history, "no gt")
        end
        if verbose
            push!(liklihood_history, ccall(host(:dpmmh_log_posterior), Cdouble, (Ptr{Cvoid},), e.model))      # calculate_posterior, :458-470
            println("Iteration: ", i, " || Clusters count: ", K, " || Log posterior: ", liklihood_history[end], " || Vi score: ", vi,
                    " || NMI score: ", nmi_score_history[end], " || Iter Time:", dt, " || Total time:", sum(iter_count))
        else
            push!(liklihood_history, 1)                        # (sic: :388)
        end
        if i % save_interval == 0 && save_model
            println("Saving Model:")
            save_checkpoint(e, save_path, save_prefix, i, time() - start_time, burnout)
        end
    end
    return iter_count, nmi_score_history, liklihood_history, cluster_count_history
end

# ---- the reference's entry points over the native engine -------------------------------------------------------------------------
# dp_parallel(all_data, local_hyper_params, α_param, iters, init_clusters, seed, verbose, save_model, burnout, gt, max_clusters,
#             outlier_weight, outlier_params, smart_splits)          src/dp-parallel-sampling.jl:121-157
# returns (dp_model, iter_count, nmi_score_history, liklihood_history, cluster_count_history); dispatches on any distribution_hyper_params
# (niw_hyperparams and multinomial_hyper are the two the engine implements: prior_kind / set_prior!)
function gpu_dp_parallel(all_data::AbstractArray{Float32,2}, local_hyper_params::distribution_hyper_params, α_param::Float32,
                         iters::Int64 = 100, init_clusters::Int64 = 1, seed = nothing, verbose = true, save_model = false,
                         burnout = 15, gt = nothing, max_clusters = Inf, outlier_weight = 0, outlier_params = nothing, smart_splits = false;
                         device = 0)
    sd = seed === nothing ? rand(UInt64) : UInt64(seed)
    e = engine_create(Matrix{Float32}(all_data), local_hyper_params, α_param; init_clusters = init_clusters, seed = sd, burnout = burnout,
                      outlier_weight = outlier_weight, outlier_params = outlier_params, smart_splits = smart_splits,
                      hard_clustering = hard_clustering, device = device)
    iter_count, nmi, lik, kh = run_model!(e, iters, 1; verbose = verbose, gt = gt, max_clusters = max_clusters, save_model = save_model,
                                          burnout = burnout)
    labels, sub = engine_labels(e)
    weights = model_get(e.model, "weights", Float32, model_K(e.model))
    return (labels = labels, labels_subcluster = sub, weights = weights, engine = e), iter_count, nmi, lik, kh
end

# the distributions of the K clusters as the reference's fit returns them ([x.cluster_params.cluster_params.distribution ...], :218)
function cluster_distributions(e::GpuEngine)
    K = model_K(e.model); D = e.D
    if e.kind == 0
        mu = model_get(e.model, "mu", Float32, D, 3K)                 # [3K][D] row-major == (D, 3K) column-major
        R  = model_get(e.model, "R", Float32, D, D, 3K)               # [3K][D][D] row-major: R[:, :, j]' is the upper-triangular factor of Σ⁻¹
        ld = model_get(e.model, "logdet", Float32, 3K)
        return [(Rk = permutedims(R[:, :, 3k - 2]); invΣ = Rk' * Rk; mv_gaussian(mu[:, 3k - 2], inv(invΣ), invΣ, ld[3k - 2], UpperTriangular(Float64.(Rk)))) for k in 1:K]   # invChol = cholesky(invΣ).U = R (priors/niw.jl:37-39)
    end
    logp = model_get(e.model, "logp", Float32, D, 3K)
    return [multinomial_dist(logp[:, 3k - 2]) for k in 1:K]
end

# fit(all_data, local_hyper_params, α_param; iters, init_clusters, seed, verbose, save_model, burnout, gt, max_clusters, outlier_weight,
#     outlier_params, smart_splits)                                src/dp-parallel-sampling.jl:215-219
# -> (labels, clusters, weights, iter_count, nmi_score_history, liklihood_history, cluster_count_history, labels_subcluster, dp_model)
function gpu_fit(all_data::AbstractArray{Float32,2}, local_hyper_params::distribution_hyper_params, α_param::Float32;
                 iters::Int64 = 100, init_clusters::Int64 = 1, seed = nothing, verbose = true, save_model = false, burnout = 20,
                 gt = nothing, max_clusters = Inf, outlier_weight = 0, outlier_params = nothing, smart_splits = false, device = 0)
    dp_model, iter_count, nmi, lik, kh = gpu_dp_parallel(all_data, local_hyper_params, α_param, iters, init_clusters, seed, verbose,
                                                         save_model, burnout, gt, max_clusters, outlier_weight, outlier_params, smart_splits;
                                                         device = device)
    return dp_model.labels, cluster_distributions(dp_model.engine), dp_model.weights, iter_count, nmi, lik, kh, dp_model.labels_subcluster, dp_model
end
# default NIW prior of fit(all_data, α_param; ...), src/dp-parallel-sampling.jl:270-276
gpu_fit(all_data::AbstractArray{Float32,2}, α_param::Float32; kw...) =
    gpu_fit(all_data, niw_hyperparams(1.0f0, zeros(Float32, size(all_data, 1)), Float32(size(all_data, 1) + 3),
                                      Matrix{Float32}(I, size(all_data, 1), size(all_data, 1))), α_param; kw...)
# the coercing methods (:279-293)
gpu_fit(all_data::AbstractArray, α_param; iters = 100, init_clusters = 1, kw...) =
    gpu_fit(Float32.(all_data), Float32(α_param); iters = Int64(iters), init_clusters = Int64(init_clusters), kw...)
gpu_fit(all_data::AbstractArray, local_hyper_params::distribution_hyper_params, α_param; iters = 100, init_clusters = 1, kw...) =
    gpu_fit(Float32.(all_data), local_hyper_params, Float32(α_param); iters = Int64(iters), init_clusters = Int64(init_clusters), kw...)
