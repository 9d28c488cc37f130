# host_transport_mpi.jl -- a caller-supplied transport for the one exchange of a sweep (dpmm_comm_init_host, include/dpmm_hip.h): hosts
# without a GPU fabric between their ranks.  What the reference's own Distributed transport does with its Dicts
# (src/local_clusters_actions.jl:231).  UNEXECUTED sketch (no Julia in the build image); needs MPI.jl.
function host_allreduce(user::Ptr{Cvoid}, buf::Ptr{Cvoid}, count::Int64, is_f64::Cint)::Cint
    a = is_f64 != 0 ? unsafe_wrap(Array, Ptr{Float64}(buf), count) : unsafe_wrap(Array, Ptr{Int64}(buf), count)
    MPI.Allreduce!(a, +, MPI.COMM_WORLD); return 0
end
const host_allreduce_c = @cfunction(host_allreduce, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint))
ccall((:dpmm_comm_init_host, libdpmmhip), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}), ctx, rank, world, host_allreduce_c, C_NULL)
