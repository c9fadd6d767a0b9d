# MultiGridBarrierHIP.jl -- Julia host side of libmgb_hip.so (include/mgb_hip.h), the MI355X-native counterpart of
# MultiGridBarrierMPI.jl's distributed path.  It mirrors, one to one, what /src/MultiGridBarrierMPI.jl does for
# HPCSparseArrays types: the ten extension hooks of MultiGridBarrier (src:57-192), `MultiGridBarrier.solve`, the
# conversions native_to_mpi / mpi_to_native (src:259-517) and the fem{1,2,3}d_mpi[_solve] wrappers (src:559-745), with
# device arrays that are thin handles into the C ABI.
#
# STATUS: written blind.  The build image has no Julia toolchain (SURVEY.md section 8c), so this file has never been LOADED by
# Julia -- lowering and dispatch are unchecked; what is checked (tests/test_host_logic.py) is that every ccall names a declared
# entry point with the declared number and kinds of arguments.  Every entry point is exercised through the identical ctypes
# binding (multigridbarriermpi.jl_amd/_lib.py) by the Python test-suite.  Keep the two bindings in step.
module MultiGridBarrierHIP

using LinearAlgebra, SparseArrays
using MultiGridBarrier
using MultiGridBarrier: Geometry, AMGBSOL, ParabolicSOL, fem1d, fem2d, fem3d, amgb, parabolic_solve
import MultiGridBarrier: amgb_zeros, amgb_all_isfinite, amgb_diag, amgb_blockdiag, map_rows, map_rows_gpu,
                         vertex_indices, _raw_array, _to_cpu_array, _rows_to_svectors

const LIB = get(ENV, "MGB_HIP_LIB", "libmgb_hip.so")
const Handle = Ptr{Cvoid}

struct MGBError <: Exception
    code::Cint
    msg::String
end
# every entry point returns a status code; the message of the last failure is thread local (mgb_last_error)
function check(rc::Cint)
    rc == 0 || throw(MGBError(rc, unsafe_string(ccall((:mgb_last_error, LIB), Cstring, ()))))
    nothing
end
macro mgb(f, argt, args...)      # @mgb <entry point> (T1, T2, ...) a1 a2 ...  ->  check(ccall(...))
    esc(:(check(ccall(($(QuoteNode(f)), LIB), Cint, $argt, $(args...)))))
end

# ---------------------------------------------------------------------------------------------- backend + array types
"One GPU + one stream: replaces the HPCBackend instance of src:84-114."
mutable struct HIPBackend
    h::Handle
    function HIPBackend(device::Integer = 0)
        r = Ref{Handle}(C_NULL)
        @mgb mgb_ctx_create (Cint, Ref{Handle}) device r
        finalizer(b -> ccall((:mgb_ctx_destroy, LIB), Cint, (Handle,), b.h), new(r[]))
    end
end
const _BACKENDS = Dict{Int,HIPBackend}()
backend_hip(device::Integer = 0) = get!(() -> HIPBackend(device), _BACKENDS, Int(device))      # cached like src:84-110

"Device fp64 vector (HPCVector: `.v`, src:175)."
mutable struct HIPVector <: AbstractVector{Float64}
    h::Handle
    n::Int
    backend::HIPBackend
end
function HIPVector(v::AbstractVector{<:Real}, b::HIPBackend = backend_hip())
    host = Vector{Float64}(v)
    r = Ref{Handle}(C_NULL)
    @mgb mgb_vec_create (Handle, Cint, Ptr{Cdouble}, Ref{Handle}) b.h length(host) host r
    finalizer(x -> ccall((:mgb_vec_free, LIB), Cint, (Handle,), x.h), HIPVector(r[], length(host), b))
end
HIPVector(n::Integer, b::HIPBackend = backend_hip()) = HIPVector(zeros(n), b)
Base.size(x::HIPVector) = (x.n,)
function Base.Vector(x::HIPVector)                                   # Vector(x) gather, src:360
    out = Vector{Float64}(undef, x.n)
    @mgb mgb_vec_download (Handle, Ptr{Cdouble}) x.h out
    out
end
Base.getindex(x::HIPVector, i::Int) = Vector(x)[i]                   # scalar indexing = host copy, as _to_cpu_array

"Device dense n x k matrix, ROW-major behind the handle (HPCMatrix: `.A`, src:176)."
struct HIPMatrix <: AbstractMatrix{Float64}
    v::HIPVector
    dims::Tuple{Int,Int}
end
HIPMatrix(A::AbstractMatrix{<:Real}, b::HIPBackend = backend_hip()) =
    HIPMatrix(HIPVector(vec(permutedims(Matrix{Float64}(A))), b), size(A))
Base.size(A::HIPMatrix) = A.dims
Base.Matrix(A::HIPMatrix) = permutedims(reshape(Vector(A.v), A.dims[2], A.dims[1]))     # Matrix(x) gather, src:357
Base.getindex(A::HIPMatrix, i::Int, j::Int) = Matrix(A)[i, j]
function Base.getindex(A::HIPMatrix, ::Colon, j::Int)                # y[:, j] -> vector on the device, test_column_extract.jl:50
    out = HIPVector(A.dims[1], A.v.backend)
    @mgb mgb_col_extract (Handle, Cint, Cint, Cint, Handle) A.v.h A.dims[1] A.dims[2] (j - 1) out.h
    out
end

"Device CSR matrix, Int32 indices (HPCSparseMatrix local block, src:216-221; Ti = Int32 as src:260)."
mutable struct HIPSparseMatrix <: AbstractMatrix{Float64}
    h::Handle
    dims::Tuple{Int,Int}
    backend::HIPBackend
end
_wrap(h::Handle, b::HIPBackend) = begin
    r, c, z = Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0)
    @mgb mgb_csr_dims (Handle, Ref{Cint}, Ref{Cint}, Ref{Cint}) h r c z
    finalizer(x -> ccall((:mgb_csr_free, LIB), Cint, (Handle,), x.h), HIPSparseMatrix(h, (Int(r[]), Int(c[])), b))
end
function HIPSparseMatrix(S::SparseMatrixCSC{<:Real}, b::HIPBackend = backend_hip())
    T = SparseMatrixCSC{Float64,Int32}(sparse(transpose(S)))          # CSC of S' == CSR of S (test_dump_matrices.jl:66-70)
    r = Ref{Handle}(C_NULL)
    @mgb mgb_csr_create (Handle, Cint, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Ref{Handle}) b.h size(S, 1) size(S, 2) (T.colptr .- Int32(1)) (T.rowval .- Int32(1)) T.nzval r
    _wrap(r[], b)
end
Base.size(A::HIPSparseMatrix) = A.dims
function SparseArrays.SparseMatrixCSC(A::HIPSparseMatrix)            # gather, src:371
    nz = Ref{Cint}(0)
    @mgb mgb_csr_dims (Handle, Ptr{Cint}, Ptr{Cint}, Ref{Cint}) A.h C_NULL C_NULL nz
    rp, ci, va = Vector{Int32}(undef, A.dims[1] + 1), Vector{Int32}(undef, nz[]), Vector{Float64}(undef, nz[])
    @mgb mgb_csr_get (Handle, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}) A.h rp ci va
    sparse(transpose(SparseMatrixCSC(A.dims[2], A.dims[1], rp .+ Int32(1), ci .+ Int32(1), va)))
end
Base.getindex(A::HIPSparseMatrix, i::Int, j::Int) = SparseMatrixCSC(A)[i, j]

# ---------------------------------------------------------------------------------------------- array algebra (SURVEY 8b)
function Base.:*(A::HIPSparseMatrix, x::HIPVector)                   # M*W, test_nonsquare.jl:43
    y = HIPVector(A.dims[1], A.backend)
    @mgb mgb_spmv (Handle, Handle, Handle) A.h x.h y.h
    y
end
_csr2(f::Symbol, A, B) = (r = Ref{Handle}(C_NULL); check(ccall((f, LIB), Cint, (Handle, Handle, Ref{Handle}), A.h, B.h, r)); _wrap(r[], A.backend))
Base.:*(A::HIPSparseMatrix, B::HIPSparseMatrix) = _csr2(:mgb_csr_spgemm, A, B)                  # M*M, test_basic_ops.jl:39
function Base.adjoint(A::HIPSparseMatrix)                            # materialised (the reference keeps a lazy Adjoint)
    r = Ref{Handle}(C_NULL)
    @mgb mgb_csr_transpose (Handle, Ref{Handle}) A.h r
    _wrap(r[], A.backend)
end
function _add(A::HIPSparseMatrix, alpha::Float64, B::HIPSparseMatrix)
    r = Ref{Handle}(C_NULL)
    @mgb mgb_csr_add (Handle, Cdouble, Handle, Ref{Handle}) A.h alpha B.h r
    _wrap(r[], A.backend)
end
Base.:+(A::HIPSparseMatrix, B::HIPSparseMatrix) = _add(A, 1.0, B)                               # test_matrix_addition.jl:48-63
Base.:-(A::HIPSparseMatrix, B::HIPSparseMatrix) = _add(A, -1.0, B)
_list(f::Symbol, Ms) = (r = Ref{Handle}(C_NULL); hs = Handle[M.h for M in Ms];
                        check(ccall((f, LIB), Cint, (Cint, Ptr{Handle}, Ref{Handle}), length(hs), hs, r)); _wrap(r[], Ms[1].backend))
Base.hcat(Ms::HIPSparseMatrix...) = _list(:mgb_csr_hcat, Ms)                                    # test_d0_construction.jl:92-100
SparseArrays.blockdiag(Ms::HIPSparseMatrix...) = _list(:mgb_csr_blockdiag, Ms)
_red(f::Symbol, x::HIPVector) = (o = Ref{Cdouble}(0); check(ccall((f, LIB), Cint, (Handle, Ref{Cdouble}), x.h, o)); o[])
LinearAlgebra.norm(x::HIPVector) = _red(:mgb_norm, x)
Base.sum(x::HIPVector) = _red(:mgb_sum, x)
function LinearAlgebra.dot(x::HIPVector, y::HIPVector)
    o = Ref{Cdouble}(0)
    @mgb mgb_dot (Handle, Handle, Ref{Cdouble}) x.h y.h o
    o[]
end
function _axpy(x::HIPVector, a::Float64, y::HIPVector)
    out = HIPVector(x.n, x.backend)
    @mgb mgb_axpy (Handle, Cdouble, Handle, Handle) x.h a y.h out.h
    out
end
Base.:+(x::HIPVector, y::HIPVector) = _axpy(x, 1.0, y)
Base.:-(x::HIPVector, y::HIPVector) = _axpy(x, -1.0, y)
function Base.Broadcast.broadcasted(::typeof(*), x::HIPVector, y::HIPVector)                    # w .* col, test_column_extract.jl:65
    out = HIPVector(x.n, x.backend)
    @mgb mgb_mul (Handle, Handle, Handle) x.h y.h out.h
    out
end

# ---------------------------------------------------------------------------------------------- the ten hooks (src:62-192)
amgb_zeros(A::HIPSparseMatrix, m, n) = HIPSparseMatrix(spzeros(m, n), A.backend)                # src:66-69
amgb_zeros(A::HIPMatrix, m, n) = HIPMatrix(zeros(m, n), A.v.backend)                            # src:72-75
amgb_zeros(::Type{HIPVector}, m) = HIPVector(m)                                                 # src:116
function amgb_all_isfinite(z::Union{HIPVector,HIPMatrix})                                       # src:121-133 (one flag back)
    o = Ref{Cint}(0)
    @mgb mgb_all_isfinite (Handle, Ref{Cint}) (z isa HIPMatrix ? z.v.h : z.h) o
    o[] != 0
end
function amgb_diag(A::Union{HIPSparseMatrix,HIPMatrix}, z::Union{HIPVector,Vector{Float64}}, m = length(z), n = length(z))   # src:137-147
    b = A isa HIPMatrix ? A.v.backend : A.backend
    zd = z isa HIPVector ? z : HIPVector(z, b)
    r = Ref{Handle}(C_NULL)
    @mgb mgb_diag (Handle, Handle, Cint, Cint, Ref{Handle}) b.h zd.h m n r
    _wrap(r[], b)
end
amgb_blockdiag(args::HIPSparseMatrix...) = blockdiag(args...)                                   # src:150
_raw_array(x::HIPVector) = x                                                                    # src:175-176
_raw_array(x::HIPMatrix) = x.v
_rows_to_svectors(M::HIPMatrix) = MultiGridBarrier._rows_to_svectors(Matrix(M))                 # src:178-181 (host rows)
_rows_to_svectors(v::HIPVector) = Vector(v)
_to_cpu_array(x::HIPMatrix) = Matrix(x)                                                         # src:183-188
_to_cpu_array(x::HIPVector) = Vector(x)
vertex_indices(A::Union{HIPVector,HIPMatrix}) = 1:size(A, 1)                                    # src:191-192

# map_rows (src:161-170).  An arbitrary Julia closure cannot cross a C ABI: it is evaluated on the host on a device->host
# copy and the result goes back up -- the trade the reference makes with _to_cpu_array (src:183-188).  The barrier family of
# the Newton hot path never comes through here: the `MultiGridBarrier.amgb(::HIPGeometry)` method at the end of this file hands
# the whole solve to the library, whose fused kernels evaluate F / F1 / F2 of `convex_Euclidian_power` (and intersections).
const AnyHIP = Union{HIPVector,HIPMatrix}
function map_rows(f, A::AnyHIP, args...)
    host = map(a -> a isa HIPMatrix ? Matrix(a) : a isa HIPVector ? Vector(a) : a, (A, args...))
    rows = [f((h isa AbstractMatrix ? view(h, i, :) : view(h, i:i) for h in host)...) for i in 1:size(host[1], 1)]
    b = A isa HIPMatrix ? A.v.backend : A.backend
    first(rows) isa Number ? HIPVector(Float64.(rows), b) : HIPMatrix(reduce(vcat, (reshape(collect(Float64, r), 1, :) for r in rows)), b)
end
map_rows_gpu(f, A::AnyHIP, args...) = map_rows(f, A, args...)
"F (which = 0), F1 (1) or F2 (2) of an intersection of power cones / half spaces, in the encoding of mgb_amg_create_terms: the
closures `map_rows` recognises and runs as fused HIP kernels (mgb_map_rows_barrier) instead of the host fallback."
struct BarrierFn
    which::Cint; K::Cint
    kind::Vector{Cint}; nq::Vector{Cint}; idx_q::Vector{Cint}; idx_s::Vector{Cint}; idx_s2::Vector{Cint}
    p::Vector{Cdouble}; coef::Vector{Cdouble}; off::Vector{Cdouble}
end
function map_rows(f::BarrierFn, ::AnyHIP, Dz::HIPMatrix)
    n, w = size(Dz, 1), (1, f.K, f.K^2)[f.which + 1]
    out = HIPVector(n * w, Dz.v.backend)
    @mgb mgb_map_rows_barrier (Cint, Cint, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Handle, Handle) f.which f.K length(f.kind) f.kind f.nq f.idx_q f.idx_s f.idx_s2 f.p f.coef f.off n Dz.v.h out.h
    f.which == 0 ? out : HIPMatrix(out, (n, w))
end

# MultiGridBarrier.solve(A, b) = A \ b (test/test_instrumented_solve.jl:25-28,99): the level's fixed-pattern device
# Cholesky lives behind an AMG handle (mgb_amg_solve_linear_gpu); a bare HIPSparseMatrix is solved on the host.
MultiGridBarrier.solve(A::HIPSparseMatrix, b::HIPVector) = HIPVector(SparseMatrixCSC(A) \ Vector(b), A.backend)

# ---------------------------------------------------------------------------------------------- conversions (src:259-517)
"native_to_mpi counterpart (src:259-338): same key order (sorted), Ti = Int32, arrays become device handles."
function native_to_hip(g::Geometry; backend::HIPBackend = backend_hip())
    up(S) = HIPSparseMatrix(SparseMatrixCSC{Float64,Int}(S), backend)
    ops = Dict{Symbol,HIPSparseMatrix}(k => up(g.operators[k]) for k in sort(collect(keys(g.operators))))
    subs = Dict{Symbol,Vector{HIPSparseMatrix}}(k => HIPSparseMatrix[up(S) for S in g.subspaces[k]] for k in sort(collect(keys(g.subspaces))))
    Geometry{Float64,HIPMatrix,HIPVector,HIPSparseMatrix,typeof(g.discretization)}(                  # explicit parameters, as src:329
        g.discretization, HIPMatrix(g.x, backend), HIPVector(g.w, backend), subs, ops,
        HIPSparseMatrix[up(S) for S in g.refine], HIPSparseMatrix[up(S) for S in g.coarsen])
end
"mpi_to_native counterpart (src:355-517): gathers geometry, AMGBSOL and ParabolicSOL back to native arrays."
hip_to_native(x::HIPVector) = Vector(x)
hip_to_native(x::HIPMatrix) = Matrix(x)
hip_to_native(x::HIPSparseMatrix) = SparseMatrixCSC(x)
hip_to_native(x::NamedTuple) = map(hip_to_native, x)                                            # SOL_main / SOL_feasibility, src:427-455
hip_to_native(x) = x
hip_to_native(g::Geometry) = Geometry(g.discretization, Matrix(g.x), Vector(g.w),
    Dict(k => SparseMatrixCSC.(v) for (k, v) in g.subspaces), Dict(k => SparseMatrixCSC(v) for (k, v) in g.operators),
    SparseMatrixCSC.(g.refine), SparseMatrixCSC.(g.coarsen))
hip_to_native(s::AMGBSOL) = AMGBSOL(hip_to_native(s.z), hip_to_native(s.SOL_feasibility), hip_to_native(s.SOL_main), s.log, hip_to_native(s.geometry))
hip_to_native(s::ParabolicSOL) = ParabolicSOL(hip_to_native(s.geometry), s.ts, hip_to_native.(s.u))

# ---------------------------------------------------------------------------------------------- whole solves on the GPU
# `amgb` / `parabolic_solve` on a Geometry whose arrays are HIP types hand the WHOLE solve to the library (the per-row hooks above
# are for callers that use the array types directly): the caller's geometry is uploaded matrix by matrix (the native_to_mpi
# input side, src:259-302), the AMG hierarchy and the barrier problem live in HBM, mgb_amg_solve runs the main phase.
const HIPGeometry{T,D} = Geometry{T,HIPMatrix,HIPVector,HIPSparseMatrix,D}
_csr0(S) = (T = SparseMatrixCSC{Float64,Int32}(sparse(transpose(SparseMatrixCSC(S)))); (T.colptr .- Int32(1), T.rowval .- Int32(1), T.nzval))
"rows per element = size of the diagonal blocks of an element-local operator (1 if it has none)"
function _block_size(S::SparseMatrixCSC)
    n, hi, start, sizes = size(S, 1), 0, 1, Int[]
    T = sparse(transpose(S))                                         # column i of T = row i of S
    for i in 1:n
        hi = max(hi, i, maximum(rowvals(T)[nzrange(T, i)]; init = i))
        hi == i && (push!(sizes, i - start + 1); start = i + 1)
    end
    all(==(sizes[1]), sizes) ? sizes[1] : 1
end
function _geo_handle(g::Geometry)
    x, w = Matrix{Float64}(hip_to_native(g.x)), Vector{Float64}(hip_to_native(g.w))
    n, dim, L = size(x, 1), size(x, 2), length(g.refine)
    dx = SparseMatrixCSC(g.operators[:dx])
    h = Ref{Handle}(C_NULL)
    @mgb mgb_geo_create (Cint, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Handle}) n dim L _block_size(dx) vec(permutedims(x)) w h
    put(name, S) = begin
        rp, ci, va = _csr0(S)
        @mgb mgb_geo_set_matrix (Handle, Cstring, Cint, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}) h[] name size(S, 1) size(S, 2) rp ci va
    end
    foreach(k -> put("op:$k", g.operators[k]), sort(collect(keys(g.operators))))                  # sorted keys, src:271-302
    for k in sort(collect(keys(g.subspaces))), (l, S) in enumerate(g.subspaces[k]); put("sub:$k:$(l - 1)", S); end
    for (l, S) in enumerate(g.refine); put("refine:$(l - 1)", S); end
    for (l, S) in enumerate(g.coarsen); put("coarsen:$(l - 1)", S); end
    h[]
end
_pairs(M) = String[string(M[i, j]) for i in 1:size(M, 1) for j in 1:2]                             # [:u :dirichlet; :s :full] -> flat strings
_rows(f, x) = permutedims(reduce(hcat, [Float64.(collect(f(x[i, :]))) for i in 1:size(x, 1)]))
"barrier terms in the encoding of mgb_amg_create_terms: cones = [(idx::Vector{Int} (1-based rows of D: q..., s), p)]"
function _amg(geo::Handle, b::HIPBackend, sv, D, cones)
    K, nt = size(D, 1), length(cones)
    kind, nq, iq, is, is2 = zeros(Cint, nt), Cint[length(c[1]) - 1 for c in cones], zeros(Cint, 3nt), Cint[c[1][end] - 1 for c in cones], fill(Cint(-1), nt)
    for (c, (idx, _)) in enumerate(cones), i in 1:length(idx)-1; iq[3(c-1)+i] = idx[i] - 1; end
    a = Ref{Handle}(C_NULL)
    @mgb mgb_amg_create_terms (Handle, Handle, Cint, Ptr{Cstring}, Cint, Ptr{Cstring}, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Handle}) b.h geo size(sv, 1) _pairs(sv) K _pairs(D) nt kind nq iq is is2 Cdouble[c[2] for c in cones] C_NULL C_NULL a
    a[]
end
function _run(a::Handle, n::Int, S::Int, c::Matrix{Float64}, z::Matrix{Float64}; tol, t, kappa, maxit, verbose, solver = 0)
    @mgb mgb_amg_set_c (Handle, Ptr{Cdouble}) a vec(permutedims(c))
    @mgb mgb_amg_set_z (Handle, Ptr{Cdouble}) a vec(z)
    @mgb mgb_amg_set_solver (Handle, Cint) a solver
    @mgb mgb_amg_prepare (Handle, Cint) a (-1)
    @mgb mgb_amg_solve (Handle, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint) a tol t kappa maxit 0 (verbose ? 1 : 0)
    zo = Vector{Float64}(undef, n * S)
    @mgb mgb_amg_get_z (Handle, Ptr{Cdouble}) a zo
    nt, te, L = Ref{Cint}(0), Ref{Cdouble}(0), Ref{Cint}(0)
    @mgb mgb_amg_sol_info (Handle, Ref{Cint}, Ref{Cdouble}, Ptr{Cdouble}, Ptr{Clonglong}) a nt te C_NULL C_NULL
    @mgb mgb_amg_dims (Handle, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ref{Cint}, Ptr{Cint}) a C_NULL C_NULL C_NULL L C_NULL
    its, ts, cd = Matrix{Clonglong}(undef, L[], nt[]), Vector{Float64}(undef, nt[]), Vector{Float64}(undef, nt[])
    @mgb mgb_amg_sol_get (Handle, Ptr{Clonglong}, Ptr{Cdouble}, Ptr{Cdouble}) a its ts cd
    reshape(zo, n, S), (t_elapsed = te[], ts = ts, its = Int.(its), c_dot_Dz = cd)                   # SOL_main fields, api.md:97-101
end
_default_D(dim) = dim == 1 ? [:u :id; :u :dx; :s :id] : dim == 2 ? [:u :id; :u :dx; :u :dy; :s :id] : [:u :id; :u :dx; :u :dy; :u :dz; :s :id]   # src:736
"MultiGridBarrier.amgb on a HIP geometry (called at src:599,666,744): returns AMGBSOL in the reference's field order (src:467-473)."
function MultiGridBarrier.amgb(geometry::HIPGeometry{T,Dsc}; p = one(T), state_variables = [:u :dirichlet; :s :full],
                               D = _default_D(size(geometry.x, 2)), f = x -> [0.5; zeros(length(x)); 1.0],
                               g = x -> length(x) == 1 ? [x[1], 2.0] : [sum(abs2, x), 100.0],                          # src:737-738
                               tol = sqrt(eps(Float64)), t = 0.1, kappa = 10.0, maxit = 10000, verbose = false, logfile = nothing,
                               solver = 0, kwargs...) where {T,Dsc}
    gfun, g = g, geometry
    dim, b, x = size(g.x, 2), g.x.v.backend, Matrix(g.x)
    n, K, S = size(x, 1), size(D, 1), size(state_variables, 1)
    idx = collect(K-dim:K)                                                                           # convex_Euclidian_power(idx = 2:dim+2)
    geo = _geo_handle(g)
    a = _amg(geo, b, state_variables, D, [(idx, Float64(p))])
    try
        c, z0 = _rows(f, x), _rows(gfun, x)
        # feasibility phase (SOL_feasibility, src:428-455), closed form for the power cone whose slack is `id` of a :full variable:
        # shift that variable by sigma = 1 + max(|q|^p - s) (Dz at s = 0 from the device)
        @mgb mgb_amg_set_c (Handle, Ptr{Cdouble}) a vec(permutedims(c))
        @mgb mgb_amg_set_z (Handle, Ptr{Cdouble}) a vec(z0)
        N = Ref{Cint}(0)
        @mgb mgb_amg_level_size (Handle, Cint, Ref{Cint}, Ptr{Cint}) a (length(g.refine) - 1) N C_NULL
        Dz = Matrix{Float64}(undef, K, n)                                                            # row-major n x K behind the ABI
        @mgb mgb_amg_apply_D (Handle, Cint, Ptr{Cdouble}, Ptr{Cdouble}) a (length(g.refine) - 1) zeros(N[]) Dz
        viol = [sqrt(sum(abs2, Dz[idx[1:end-1], i]))^p - Dz[idx[end], i] for i in 1:n]
        feas = nothing
        if maximum(viol) >= 0
            sigma = 1 + maximum(viol)
            svar = findfirst(==(D[idx[end], 1]), state_variables[:, 1])
            (D[idx[end], 2] == :id && state_variables[svar, 2] == :full) || error("amgb: feasibility phase needs the cone's slack to be `id` of a :full variable")
            z0[:, svar] .+= sigma
            feas = (shift = sigma, ts = Float64[], its = zeros(Int, length(g.refine), 0), c_dot_Dz = Float64[], t_elapsed = 0.0)
        end
        z, main = _run(a, n, S, c, z0; tol = tol, t = t, kappa = kappa, maxit = maxit, verbose = verbose, solver = solver)
        zh = HIPMatrix(z, b)
        AMGBSOL{T,typeof(zh),typeof(g.w),HIPSparseMatrix,Dsc}(zh, feas, main, String[], g)           # src:467-473
    finally
        ccall((:mgb_amg_destroy, LIB), Cint, (Handle,), a)
        ccall((:mgb_geo_destroy, LIB), Cint, (Handle,), geo)
    end
end
"MultiGridBarrier.parabolic_solve on a HIP geometry (src:22,54; ParabolicSOL src:512-516): implicit Euler, per step the two-cone problem s1 >= u^2, s2 >= |grad u|^p."
function MultiGridBarrier.parabolic_solve(geometry::HIPGeometry{T,Dsc}; h = 0.2, t0 = 0.0, t1 = 1.0, p = one(T), f1 = x -> 0.5,
                                          g = x -> length(x) == 1 ? [x[1], 2.0] : [sum(abs2, x), 100.0], tol = sqrt(eps(Float64)),
                                          verbose = false, kwargs...) where {T,Dsc}
    gfun, g = g, geometry
    dim, b, x = size(g.x, 2), g.x.v.backend, Matrix(g.x)
    n, K = size(x, 1), dim + 3
    sv = [:u :dirichlet; :s1 :full; :s2 :full]
    D = vcat([:u :id], reduce(vcat, [[:u o] for o in (:dx, :dy, :dz)[1:dim]]), [:s1 :id; :s2 :id])
    geo = _geo_handle(g)
    a = _amg(geo, b, sv, D, [([1, K - 1], 2.0), (vcat(collect(2:dim+1), K), Float64(p))])
    try
        u0 = [Float64(gfun(x[i, :])[1]) for i in 1:n]
        grad2 = sum(abs2.(SparseMatrixCSC(g.operators[o]) * u0) for o in (:dx, :dy, :dz)[1:dim])
        z = hcat(u0, fill(1 + maximum(abs2, u0), n), fill(1 + maximum(grad2 .^ (p / 2)), n))
        fg = [Float64(f1(x[i, :])) for i in 1:n]
        ts = collect(t0:h:t1)
        u = [HIPMatrix(z, b)]
        for _ in 2:length(ts)
            c = zeros(n, K)
            c[:, 1] = fg .- z[:, 1] ./ h
            c[:, K-1] .= 1 / (2h)
            c[:, K] .= 1 / p
            z, _ = _run(a, n, 3, c, z; tol = tol, t = 0.1, kappa = 10.0, maxit = 10000, verbose = verbose)
            push!(u, HIPMatrix(z, b))
        end
        ParabolicSOL(g, ts, u)                                                                       # src:512-516
    finally
        ccall((:mgb_amg_destroy, LIB), Cint, (Handle,), a)
        ccall((:mgb_geo_destroy, LIB), Cint, (Handle,), geo)
    end
end
# the reference's entry-point names (src:559,594,626,661,696,735) with backend = backend_hip(): kwargs go to both callees
for (mk, sol, fem) in ((:fem1d_mpi, :fem1d_mpi_solve, :fem1d), (:fem2d_mpi, :fem2d_mpi_solve, :fem2d), (:fem3d_mpi, :fem3d_mpi_solve, :fem3d))
    @eval $mk(::Type{T} = Float64; backend::HIPBackend = backend_hip(), kwargs...) where {T} = native_to_hip($fem(T; kwargs...); backend = backend)
    @eval $sol(::Type{T} = Float64; kwargs...) where {T} = MultiGridBarrier.amgb($mk(T; kwargs...); kwargs...)
end

export HIPBackend, backend_hip, HIPVector, HIPMatrix, HIPSparseMatrix, HIPGeometry, native_to_hip, hip_to_native,
       fem1d_mpi, fem2d_mpi, fem3d_mpi, fem1d_mpi_solve, fem2d_mpi_solve, fem3d_mpi_solve, MGBError
end # module
