# DSMGPHip.jl -- reference-side binding of the MI355X GP-expert path (libdsmgp_hip.so, C ABI: include/dsmgp_hip.h).
#
# What a maintainer of trappmartin/DeepStructuredMixtures adds next to `src/` to run the hot path on the GPU:
#
#     using DeepStructuredMixtures
#     include("julia/DSMGPHip.jl"); using .DSMGPHip
#     model = buildDSMGP(x, y, 3, 4; M = 200, kernel = IsoSE(log(0.3), 0.0), logNoise = log(0.1))   # host build, as before
#     DSMGPHip.attach!(model)                 # upload X, y and the leaf table once; from here on the methods below run
#     fit!(model); update!(model); μ, σ² = predict(model, xtest); train!(model, ADAM())
#
# The file re-defines exactly the methods whose bodies are the hot path (SURVEY.md section 8(b)); everything else --
# tree construction, routing (`getchild`), `update!`/`infer!`, the `train!`/`finetune!` loops, Flux -- runs as it is:
#
#   fit!(spn, D, gpmap; τ)          src/fit.jl:71-122      schedule decided here as there, executed by dsmgp_set_sharing + dsmgp_fit
#   fit_naive!(spn)                 src/fit.jl:294-304     dsmgp_set_sharing(NULL) + dsmgp_fit
#   update_cholesky!(gp)            src/gaussianprocess.jl:82-108   one-leaf session (a GP that is not a leaf of an attached model)
#   mll(gp)                         src/gaussianprocess.jl:163      per-leaf value returned by dsmgp_fit
#   prediction(gp, xtest)           src/gaussianprocess.jl:131-137  dsmgp_predict_leaves on the GP's session
#   predict(model, x)               src/common.jl:294-307  DSMGP: rows routed on the device (dsmgp_set_tree once per attach!,
#                                                          dsmgp_set_test_routed per test set), ONE dsmgp_predict_run +
#                                                          dsmgp_aggregate; PoE family: every row to every expert
#   updategradients!(spn)           src/fit.jl:306-311     dsmgp_gradients, results written to kernel.∂ℓ/∂σ and gp.∂ϵ
#   ∇mll(gp)                        src/gaussianprocess.jl:185-190  reads those fields (no second updategradients!)
#   setparams!(spn, hyp)            src/optimize.jl:188-198 unchanged on the host; fit! pushes the vectors (dsmgp_set_hyper)
#
# Julia is not installed in the image this repository is built in: the file has been written against the reference's
# sources and the header, line by line, and NOT executed.  The same entry points in the same order are exercised by
# tests/c_abi_smoke.c (plain C) and by the Python mirror (deepstructuredmixtures_amd/hipabi.py, model.py).
module DSMGPHip

using LinearAlgebra
using Libdl
using DeepStructuredMixtures
import DeepStructuredMixtures: fit!, fit_naive!, update_cholesky!, prediction, mll, predict, updategradients!, ∇mll
using DeepStructuredMixtures: GPNode, GPSumNode, GPSplitNode, DSMGP, PoE, gPoE, rBCM, BiDict, GaussianProcess,
                              IsoSE, ArdSE, IsoLinear, ConstMean, getLeaves, getchild, children, logweights, getnoise

export attach!, detach!, census

# ---------------------------------------------------------------------------------------------- library
const LIB = Ref{Ptr{Cvoid}}(C_NULL)
function lib()
    if LIB[] == C_NULL
        LIB[] = dlopen(get(ENV, "DSMGP_HIP_LIB", "libdsmgp_hip.so"))
    end
    return LIB[]
end
sym(s::Symbol) = dlsym(lib(), s)

const SHARE_FULL, SHARE_COPY, SHARE_PREFIX = Int32(0), Int32(1), Int32(2)      # DSMGP_SHARE_*
const AGG_MIXTURE, AGG_POE, AGG_GPOE, AGG_RBCM = Int32(0), Int32(1), Int32(2), Int32(3)   # DSMGP_AGG_*

kind(::IsoSE) = Int32(0)
kind(::ArdSE) = Int32(1)
kind(::IsoLinear) = Int32(2)
# hyper-vector of one kernel id on the reference's log scale, [logℓ..., logσ, logNoise] (src/gaussianprocess.jl:141-161)
loghyp(k::IsoSE, ln) = Float64[k.logℓ, k.logσ, ln]
loghyp(k::ArdSE, ln) = Float64[k.logℓ..., k.logσ, ln]
loghyp(k::IsoLinear, ln) = Float64[k.logℓ, 0.0, ln]             # the variance slot is a dummy (src/kernels.jl:181-183)

# ---------------------------------------------------------------------------------------------- session
"One device context + the leaf table of one model (or of one stand-alone GaussianProcess)."
mutable struct Session
    h::Ptr{Cvoid}
    leaves::Vector{GPNode}                 # leaf l of the ABI (0-based there) = leaves[l + 1]: order of getLeaves(root)
    gps::Vector{GaussianProcess}           # leaves[l].dist, or the single GP
    index::IdDict{Any,Int}                 # GaussianProcess -> position in gps
    leafmll::Vector{Float64}
    info::Vector{Int32}
    grad::Matrix{Float64}                  # stride x L, column l = [∂ℓ..., ∂σ, ∂ϵ] of leaf l (src/gaussianprocess.jl:212-214)
    stride::Int
    testkey::UInt                          # hash of the registered test set (dsmgp_set_test is called once per test set)
    testptr::Vector{Int64}
    census::Dict{Symbol,Any}
    fitted::Bool
end

const SESSIONS = IdDict{Any,Session}()     # root node (or GaussianProcess) -> session
const OWNER = IdDict{Any,Session}()        # GaussianProcess -> the session whose leaf it is

lasterror(h) = unsafe_string(ccall(sym(:dsmgp_last_error), Cstring, (Ptr{Cvoid},), h))
chk(s::Session, rc) = rc == 0 ? nothing : error("dsmgp error $rc: " * lasterror(s.h))

function newsession(device::Integer)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall(sym(:dsmgp_create), Cint, (Int32, Ref{Ptr{Cvoid}}), Int32(device), r)
    rc == 0 || error("dsmgp_create: " * lasterror(C_NULL))        # no GPU: the path has no CPU fallback
    s = Session(r[], GPNode[], GaussianProcess[], IdDict{Any,Int}(), Float64[], Int32[], zeros(0, 0), 0, UInt(0), Int64[],
                Dict{Symbol,Any}(), false)
    finalizer(x -> (x.h != C_NULL && ccall(sym(:dsmgp_destroy), Cint, (Ptr{Cvoid},), x.h); x.h = C_NULL), s)
    return s
end

"Upload the training data and the leaf table (dsmgp_set_train / dsmgp_set_leaves).  gp.x are views of the rows of the
training matrix and gp.y is mean-subtracted (src/gaussianprocess.jl:72-74), so X and y are rebuilt from the leaves."
function upload!(s::Session, gps::Vector, obs::Vector{Vector{Int}}, kernelids::Vector{Int})
    N = maximum(maximum, obs)
    Dm = size(gps[1].x, 2)
    X = zeros(Float64, N, Dm)
    y = zeros(Float64, N)
    for (gp, o) in zip(gps, obs)
        X[o, :] = gp.x
        y[o] = gp.y .+ gp.mean.m
    end
    GC.@preserve X y chk(s, ccall(sym(:dsmgp_set_train), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int32),
                                  s.h, X, y, N, Dm))
    ptr = Int64[0]
    for o in obs
        push!(ptr, ptr[end] + length(o))
    end
    idx = Int64.(reduce(vcat, obs)) .- 1                         # 1-based in Julia, 0-based in the ABI; ascending per leaf
    kid = Int32.(kernelids .- 1)
    m = Float64[gp.mean.m for gp in gps]
    GC.@preserve ptr idx kid m chk(s, ccall(sym(:dsmgp_set_leaves), Cint,
        (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Int32}, Ptr{Float64}), s.h, length(gps), ptr, idx, kid, m))
    s.gps = collect(gps)
    empty!(s.index)
    for (l, gp) in enumerate(gps)
        s.index[gp] = l
        OWNER[gp] = s
    end
    s.leafmll = fill(NaN, length(gps))
    s.info = zeros(Int32, length(gps))
    s.stride = maximum(sum(DeepStructuredMixtures.nparams(gp)) for gp in gps)
    s.grad = zeros(s.stride, length(gps))
    s.testkey = UInt(0)
    s.fitted = false
    return s
end

"attach!(model; device = 0): route the hot-path methods of `model` to the GPU.  Returns the session."
function attach!(model::Union{DSMGP,PoE,gPoE,rBCM}; device::Integer = 0)
    return attach!(model.root; device = device)
end
function attach!(root::Union{GPSumNode,GPSplitNode}; device::Integer = 0)
    s = newsession(device)
    s.leaves = getLeaves(root)
    upload!(s, [l.dist for l in s.leaves], [l.obs for l in s.leaves], [l.kernelid for l in s.leaves])
    settree!(s, root)
    SESSIONS[root] = s
    return s
end

"The tree as the flat arrays dsmgp_set_tree takes (breadth-first, the children of a node consecutive, 0-based indices): kind
0 region / 1 split / 2 sum, first child, number of children, split dimension, ascending thresholds (column i of a thr_ld x
n_nodes matrix = node i, the last one its upper bound), leaf index of a region in the session's leaf table.  After it the
routing of predict (getchild, src/common.jl:101-122, walked per node in :181-196,275-292) runs on the device."
function settree!(s::Session, root)
    nodes = Any[root]
    kind = Int8[]; first = Int64[]; nchild = Int64[]; sdim = Int64[]; leaf = Int64[]; thr = Vector{Float64}[]
    leafidx = Dict(l.id => i - 1 for (i, l) in enumerate(s.leaves))
    i = 1
    while i <= length(nodes)
        nd = nodes[i]
        if nd isa GPNode
            push!(kind, Int8(0)); push!(first, 0); push!(nchild, 0); push!(sdim, 0); push!(leaf, leafidx[nd.id]); push!(thr, Float64[])
        else
            ch = children(nd)
            push!(kind, nd isa GPSplitNode ? Int8(1) : Int8(2))
            push!(first, length(nodes))            # 0-based index of the first child: it is appended next
            push!(nchild, length(ch)); push!(leaf, -1)
            push!(sdim, nd isa GPSplitNode ? nd.split[1][1] - 1 : 0)
            push!(thr, nd isa GPSplitNode ? Float64[t for (_, t) in nd.split] : Float64[])
            append!(nodes, ch)
        end
        i += 1
    end
    ld = max(1, maximum(length, thr))
    T = fill(Inf, ld, length(nodes))
    for (j, t) in enumerate(thr)
        T[1:length(t), j] = t
    end
    GC.@preserve kind first nchild sdim T leaf chk(s, ccall(sym(:dsmgp_set_tree), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Int8}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Int64}),
        s.h, length(nodes), kind, first, nchild, sdim, T, ld, leaf))
end
function detach!(root)
    s = pop!(SESSIONS, root, nothing)
    if s !== nothing
        foreach(gp -> delete!(OWNER, gp), s.gps)
        finalize(s)
    end
end
session(root) = get(() -> error("DSMGPHip: call attach!(model) first"), SESSIONS, root)

"Session of a GaussianProcess that is not a leaf of an attached model: a one-leaf table, created on first use."
function session(gp::GaussianProcess)
    haskey(OWNER, gp) && return OWNER[gp]
    s = newsession(0)
    upload!(s, [gp], [collect(1:gp.N)], [1])
    SESSIONS[gp] = s
    return s
end

"dsmgp_set_hyper for every kernel id, from the leaves' current fields: setparams!(spn, hyp) (src/optimize.jl:188-198) gives
every leaf of an id the same vector, so the first leaf of each id speaks for it."
function pushhyper!(s::Session, kernelids)
    seen = Set{Int}()
    for (gp, id) in zip(s.gps, kernelids)
        id in seen && continue
        push!(seen, id)
        h = loghyp(gp.kernel, gp.logNoise.value)
        GC.@preserve h chk(s, ccall(sym(:dsmgp_set_hyper), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Int32),
                                    s.h, Int32(id - 1), kind(gp.kernel), h, Int32(length(h))))
    end
end

function runfit!(s::Session)
    L = length(s.gps)
    sec = Ref{Float64}(0.0)
    mllv, info = s.leafmll, s.info
    GC.@preserve mllv info chk(s, ccall(sym(:dsmgp_fit), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Ref{Float64}),
                                        s.h, mllv, info, sec))
    bad = findfirst(!=(0), info)
    bad === nothing || throw(PosDefException(Int(info[bad])))   # LAPACK's info, which src/gaussianprocess.jl:101 ignores
    s.fitted = true
    return sec[]
end

# ---------------------------------------------------------------------------------------------- fit!
"""
    fit!(spn, D, gpmap; τ = 0.05)      (src/fit.jl:71-122)

The decisions of the reference's loop -- main leaf `argmax(D[:,j] .* D[j,:])`, processing order by `counts`, the arm of
`fitcontained!` -- are taken here exactly as there; instead of factorising leaf by leaf they are recorded as
`(op, src, plen)` and executed in one batched call.  COPY (`:132-143`) and PREFIX (`chol_continue!`, `:276-278`) are
exact and kept; the row-deletion arm (`:174-201`) is numerically defective in the reference (SURVEY F4) and becomes a
full factorisation -- `census(spn)[:lowrank_leaves]` lists the leaves it would have taken.  Returns seconds, as `@elapsed`
does there.
"""
function fit!(spn::Union{GPSumNode,GPSplitNode}, D::Matrix, gpmap::BiDict; τ = 0.05)
    s = session(spn)
    leaves = s.leaves                                           # fixed order: leaf l of the device table
    n = length(leaves)
    pos = Dict(l.id => i for (i, l) in enumerate(leaves))       # node id -> device leaf (1-based)
    counts = zeros(Int, n)
    S = Vector{Int}(undef, n)                                   # main leaf of j, in gpmap numbering
    for j in 1:n
        i = argmax(D[:, j] .* D[j, :])                          # :79
        counts[i] += 1
        S[j] = i
    end
    order = sort(collect(1:n), by = j -> counts[j])             # :86 (gpmap numbering; sort is stable like sort!)
    node(j) = leaves[pos[gpmap.fx[j]]]
    op = fill(SHARE_FULL, n); src = fill(Int32(-1), n); plen = zeros(Int64, n)      # indexed by DEVICE leaf
    arm = fill(:full, n)
    processed = falses(n)
    for j in order
        processed[j] && continue
        i = S[j]
        processed[i] = true                                     # :97-100: the main leaf is factorised in full
        processed[j] = true
        i == j && continue
        jn, mn = node(j), node(i)
        dj, di = pos[jn.id], pos[mn.id]
        (mn.kernelid != jn.kernelid || first(jn.obs) < first(mn.obs)) && continue          # :107-112
        ione = D[i, j] == one(eltype(D))
        jone = D[j, i] == one(eltype(D))
        if ione && jone                                         # identical observation sets (:132-143)
            arm[dj] = :copy
            if op[di] == SHARE_FULL
                op[dj], src[dj] = SHARE_COPY, Int32(di - 1)
            elseif op[di] == SHARE_COPY
                op[dj], src[dj] = SHARE_COPY, src[di]
            end
        elseif ione && !jone && τ > 0                           # j contains the main leaf (:208-292)
            p = mn.nobs
            if first(jn.obs) == first(mn.obs) && jn.nobs > p && jn.obs[1:p] == mn.obs
                arm[dj] = :prefix
                if op[di] == SHARE_FULL
                    op[dj], src[dj], plen[dj] = SHARE_PREFIX, Int32(di - 1), p
                end
            end
        elseif jone && !ione                                    # j is contained in the main leaf (:145-206)
            e = findfirst(==(last(jn.obs)), mn.obs)             # :168
            ndel = e - jn.nobs                                  # |setdiff(mainNode.obs[1:e], jNode.obs)| (:170)
            if ndel / jn.nobs < τ                               # :173
                arm[dj] = ndel > 0 ? :lowrank_as_full : :leading_as_full
            end
        end
    end
    s.census = Dict{Symbol,Any}(:full => count(==(:full), arm), :copy => count(==(:copy), arm),
                                :prefix => count(==(:prefix), arm), :lowrank_as_full => count(==(:lowrank_as_full), arm),
                                :leading_as_full => count(==(:leading_as_full), arm),
                                :lowrank_leaves => [leaves[l].id for l in findall(==(:lowrank_as_full), arm)],
                                :leading_leaves => [leaves[l].id for l in findall(==(:leading_as_full), arm)])
    pushhyper!(s, [l.kernelid for l in leaves])
    GC.@preserve op src plen chk(s, ccall(sym(:dsmgp_set_sharing), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}),
                                          s.h, op, src, plen))
    return runfit!(s)
end

"Census of the last fit!: counts of the reference's arms and the ids of the leaves computed in full instead."
census(spn) = session(spn).census

function fit_naive!(spn::Union{GPSplitNode,GPSumNode})          # src/fit.jl:294-304
    s = session(spn)
    pushhyper!(s, [l.kernelid for l in s.leaves])
    chk(s, ccall(sym(:dsmgp_set_sharing), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}), s.h, C_NULL, C_NULL, C_NULL))
    return runfit!(s)
end

# ---------------------------------------------------------------------------------------------- single GP
"update_cholesky!(gp) (src/gaussianprocess.jl:82-108).  A leaf of an attached model is refitted with its whole table
(one batched call: that is what fit! does there, leaf by leaf); any other GP has a one-leaf session of its own."
function update_cholesky!(gp::GaussianProcess)
    s = session(gp)
    pushhyper!(s, isempty(s.leaves) ? [1] : [l.kernelid for l in s.leaves])
    isempty(s.leaves) && chk(s, ccall(sym(:dsmgp_set_sharing), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}),
                                      s.h, C_NULL, C_NULL, C_NULL))
    runfit!(s)
    return gp
end

"mll(gp) (src/gaussianprocess.jl:163): the per-leaf value dsmgp_fit returned (−(y·α + logdet + N log 2π)/2 with y·α = z·z)."
function mll(gp::GaussianProcess)
    s = session(gp)
    s.fitted || update_cholesky!(gp)
    return s.leafmll[s.index[gp]]
end

"prediction(gp, xtest) (src/gaussianprocess.jl:131-137) -> (μ, Σ).  Only diag(Σ) is ever consumed (src/common.jl:136,147);
Σ comes back as a Diagonal (K_tt and VᵀV are never formed), noise added, no ϵ, no clamp (that is the caller's, :137)."
function prediction(gp::GaussianProcess, xtest::AbstractMatrix)
    s = session(gp)
    s.fitted || update_cholesky!(gp)
    l = s.index[gp]
    xt = Matrix{Float64}(xtest)
    nt = size(xt, 1)
    ptr = zeros(Int64, length(s.gps) + 1)
    ptr[(l + 1):end] .= nt                                      # only leaf l predicts
    idx = Int64.(0:(nt - 1))
    μ = Vector{Float64}(undef, nt); σ² = similar(μ)
    GC.@preserve xt ptr idx μ σ² chk(s, ccall(sym(:dsmgp_predict_leaves), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
        s.h, xt, nt, size(xt, 2), ptr, idx, μ, σ²))
    s.testkey = UInt(0)
    return μ, Diagonal(σ²)
end

# ---------------------------------------------------------------------------------------------- predict
"Rows of `x` every leaf is asked to predict: sums forward all rows to all children, splits one child per row
(src/common.jl:181-196,275-292 with getchild :101-122)."
function route!(rows::Dict{Symbol,Vector{Int}}, node::GPNode, x, sel::Vector{Int})
    rows[node.id] = sel
end
function route!(rows, node::GPSumNode, x, sel)
    foreach(c -> route!(rows, c, x, sel), children(node))
end
function route!(rows, node::GPSplitNode, x, sel)
    idx = getchild(node, x[sel, :])
    for (k, c) in enumerate(children(node))
        route!(rows, c, x, sel[findall(idx .== k)])
    end
end
function routeall!(rows, node::GPNode, x, sel)                  # PoE family: every expert predicts every row (:198-208)
    rows[node.id] = sel
end
routeall!(rows, node, x, sel) = foreach(c -> routeall!(rows, c, x, sel), children(node))

"Product of the sum-node weights on every leaf's path: the nested log-domain recursion of _predict (src/common.jl:275-302)
is linear in (μ, μ², σ²), i.e. the flat mixture over the visited leaves with these weights."
function pathweights!(w::Dict{Symbol,Float64}, node::GPNode, acc::Float64)
    w[node.id] = exp(acc)
end
function pathweights!(w, node::GPSumNode, acc)
    for (k, c) in enumerate(children(node))
        pathweights!(w, c, acc + logweights(node)[k])
    end
end
pathweights!(w, node::GPSplitNode, acc) = foreach(c -> pathweights!(w, c, acc), children(node))

"Register the test rows once per test set (dsmgp_set_test); a following fit! then carries them through the factorisation."
function settest!(s::Session, x::Matrix{Float64}, rows::Dict{Symbol,Vector{Int}})
    key = hash((size(x), x, [rows[l.id] for l in s.leaves]))
    key == s.testkey && return
    ptr = Int64[0]
    for l in s.leaves
        push!(ptr, ptr[end] + length(rows[l.id]))
    end
    idx = Int64.(reduce(vcat, [rows[l.id] for l in s.leaves])) .- 1
    isempty(idx) && push!(idx, 0)
    GC.@preserve x ptr idx chk(s, ccall(sym(:dsmgp_set_test), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int32, Ptr{Int64}, Ptr{Int64}),
                                        s.h, x, size(x, 1), size(x, 2), ptr, idx))
    s.testkey = key
    s.testptr = ptr
end

"Register the test rows of a DSMGP and let the device route them (dsmgp_set_test_routed: one thread per row walks the tree
handed over by settree!); only the per-leaf row offsets come back."
function settestrouted!(s::Session, x::Matrix{Float64})
    key = hash((size(x), x, :routed))
    key == s.testkey && return
    GC.@preserve x chk(s, ccall(sym(:dsmgp_set_test_routed), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int32), s.h, x, size(x, 1), size(x, 2)))
    ptr = Vector{Int64}(undef, length(s.leaves) + 1)
    GC.@preserve ptr chk(s, ccall(sym(:dsmgp_routes), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), s.h, ptr, C_NULL))
    s.testkey = key
    s.testptr = ptr
end

function aggregate(s::Session, family::Int32, coef, group, G::Integer, plain::Bool, priorkid::Integer, nt::Integer)
    sec = Ref{Float64}(0.0)
    chk(s, ccall(sym(:dsmgp_predict_run), Cint, (Ptr{Cvoid}, Ref{Float64}), s.h, sec))
    μ = Vector{Float64}(undef, nt); σ² = similar(μ)
    c = coef === nothing ? C_NULL : pointer(coef)
    g = group === nothing ? C_NULL : pointer(group)
    GC.@preserve coef group μ σ² chk(s, ccall(sym(:dsmgp_aggregate), Cint,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Int32}, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}),
        s.h, family, c, g, Int32(G), Int32(plain), Int32(priorkid), μ, σ²))
    return μ, σ²
end

"Do the weights of every sum node add up to one?  (Every path of the reference keeps them so; `logweights` is a plain field.)"
weightsnormalised(node::GPNode) = true
weightsnormalised(node::GPSplitNode) = all(weightsnormalised, children(node))
weightsnormalised(node::GPSumNode) = abs(sum(exp.(logweights(node))) - 1) <= 1e-12 && all(weightsnormalised, children(node))

"predict(model::DSMGP, x) (src/common.jl:294-304): (μ, σ²) of the mixture; σ² ≤ 0 of a leaf → ϵ inside the aggregation (:137)."
function predict(model::DSMGP, x::AbstractMatrix)
    # `_predict` shifts the means by μmin - 1 before it weighs them (src/common.jl:134-143,275-302): with weights that add up to
    # one the recursion IS the flat mixture the device aggregates; with hand-assigned ones it is not, and the reference's own
    # recursion runs (its leaf predictions come from the device through prediction(gp, x) above)
    size(x, 1) == 0 && return Float64[], Float64[]            # (dsmgp_set_test* refuse n_t = 0)
    weightsnormalised(model.root) || return predict(model.root, x)
    s = session(model.root)
    xt = Matrix{Float64}(x)
    settestrouted!(s, xt)          # (route! + settest! above are the host form of the same lists: rows in ascending order per leaf)
    w = Dict{Symbol,Float64}()
    pathweights!(w, model.root, 0.0)
    coef = Float64[w[l.id] for l in s.leaves]
    return aggregate(s, AGG_MIXTURE, coef, nothing, 0, model.root isa GPNode, 0, size(xt, 1))
end

function predictfamily(model, x::AbstractMatrix, family::Int32)
    size(x, 1) == 0 && return Float64[], Float64[]
    s = session(model.root)
    xt = Matrix{Float64}(x)
    rows = Dict{Symbol,Vector{Int}}()
    routeall!(rows, model.root, xt, collect(1:size(xt, 1)))
    settest!(s, xt, rows)
    L = length(s.leaves)
    if family == AGG_RBCM                                       # per root child (src/common.jl:224-241); prior of leftGP (:227)
        group = zeros(Int32, L)
        for (g, c) in enumerate(children(model.root)), l in getLeaves(c)
            group[s.index[l.dist]] = Int32(g - 1)
        end
        return aggregate(s, family, nothing, group, length(children(model.root)), false, s.leaves[1].kernelid - 1, size(xt, 1))
    end
    β = family == AGG_GPOE ? 1.0 / length(children(model.root)) : 1.0        # :215
    return aggregate(s, family, fill(β, L), nothing, 0, false, 0, size(xt, 1))
end
predict(model::PoE, x::AbstractMatrix) = predictfamily(model, x, AGG_POE)      # src/common.jl:305
predict(model::gPoE, x::AbstractMatrix) = predictfamily(model, x, AGG_GPOE)    # :306
predict(model::rBCM, x::AbstractMatrix) = predictfamily(model, x, AGG_RBCM)    # :307

# ---------------------------------------------------------------------------------------------- gradients
"""
    updategradients!(spn)      (src/fit.jl:306-311)

One batched dsmgp_gradients call (L⁻ᵀ by blocked triangular inversion + the contraction tiles: ≈ 2× the Cholesky flops
instead of the ≈ 36× of src/gaussianprocess.jl:165-178 + src/kernels.jl:85-99); the results land where the reference
keeps them -- `kernel.∂ℓ`, `kernel.∂σ`, `gp.∂ϵ.value` -- with its scaling (IsoSE gradients carry the extra factor σ,
ArdSE length-scale gradients are identically zero: SURVEY F6/F7).
"""
function updategradients!(spn::Union{GPSumNode,GPSplitNode})
    fetchgradients!(session(spn))
end
function updategradients!(gp::GaussianProcess)
    fetchgradients!(session(gp))
end
function fetchgradients!(s::Session)
    g = s.grad
    GC.@preserve g chk(s, ccall(sym(:dsmgp_gradients), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), s.h, g, Int32(s.stride)))
    for (l, gp) in enumerate(s.gps)
        k = gp.kernel
        nl = k isa ArdSE ? length(k.logℓ) : 1
        if k isa ArdSE
            k.∂ℓ[:] = g[1:nl, l]
        else
            k.∂ℓ = g[1, l]
        end
        k isa IsoLinear || (k.∂σ = g[nl + 1, l])
        gp.∂ϵ.value = g[nl + 2, l]
    end
    return nothing
end

"∇mll(gp) (src/gaussianprocess.jl:185-190) = [∂ℓ..., ∂σ, ∂ϵ].  The reference calls updategradients!(gp) again here (a
second 6n³ per leaf and iteration, src/optimize.jl:49); the values of the batched call are still valid: read them."
function ∇mll(gp::GaussianProcess)
    s = session(gp)
    l = s.index[gp]
    n = sum(DeepStructuredMixtures.nparams(gp))
    return s.grad[1:n, l]
end

end # module
