# README Example 1 (README.md:23-57 of tkoolen/Parametron.jl) through the reference's own update! and through the HIP backend, on the
# same Parameter values: every MOI array the optimizer would be given must agree — indices exactly, coefficients bit for bit (the
# objective is literal at this size).  Run by tests/test_gpu_julia.py when `julia` (with Parametron and MathOptInterface installed)
# exists on the GPU box:   PARAMETRON_HIP_LIB=.../libparametron_hip.so julia julia/example1_parity.jl
using Parametron, Random, LinearAlgebra
import MathOptInterface
const MOI = MathOptInterface
include(joinpath(@__DIR__, "ParametronHIP.jl"))
include(joinpath(@__DIR__, "ParametronHIPBackend.jl"))
using .ParametronHIPBackend

function build(seed)
    rng = MersenneTwister(seed)
    model = Parametron.mock_model()
    n, m = 8, 2
    x = [Variable(model) for _ = 1 : n]
    A = Parameter(a -> rand!(rng, a), zeros(n, n), model)
    b = Parameter(v -> rand!(rng, v), zeros(n), model)
    C = Parameter(c -> rand!(rng, c), zeros(m, n), model)
    d = Parameter(zeros(m), model) do d
        rand!(rng, d)
        d .*= 2
    end
    residual = @expression A * x - b
    @objective(model, Minimize, residual ⋅ residual)
    @constraint(model, C * x == d)
    model
end

same(a, b) = length(a) == length(b) && all(reinterpret(UInt8, a) .== reinterpret(UInt8, b))

ref, dev = build(1234), build(1234)             # two models, identical callback streams
Parametron.initialize!(ref)
hm = HIPModel(dev)
ok = true
for it = 1 : 3
    Parametron.update!(ref)
    Parametron.update!(hm)
    fr, fd = ref.objective.f, dev.objective.f
    global ok &= same(fr.quadratic_terms, fd.quadratic_terms) && same(fr.affine_terms, fd.affine_terms) && fr.constant === fd.constant
    cr = first(ParametronHIPBackend.constraint_records(ref)); cd = first(ParametronHIPBackend.constraint_records(dev))
    global ok &= same(cr.f.terms, cd.f.terms) && same(cr.f.constants, cd.f.constants)
end
println(ok ? "JULIA_PARITY_OK" : "JULIA_PARITY_MISMATCH")
exit(ok ? 0 : 1)
