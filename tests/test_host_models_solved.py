"""CPU: constant (Parameter-free) models solved end to end through the host API with the dense test solver — the reference's
remaining closed-form model tests.  No device is involved: constant functions are built once on the host, exactly as in the
reference (isconstant, src/moi_interop.jl:123,132,169)."""
import numpy as np
import pytest

import parametron_jl_amd as P
from parametron_jl_amd import Variable
from qp_solver import DenseQPOptimizer


def test_scalar_constraints_1_and_issue_30():                              # test/model.jl:269-279, 173-196
    for build in ("scalar", "vector", "expression"):
        model = P.Model(DenseQPOptimizer())
        x = Variable(model)
        if build == "scalar":
            P.constraint(model, x <= -3)
        elif build == "vector":
            P.constraint(model, [x], "<=", [-3.0])
        else:
            P.constraint(model, P.lazy("-", [x], [-3.0]), "<=", [0.0])    # constraintexpr(x, ub) = @expression [x] - [ub]
        P.objective(model, P.Minimize, x ** 2)
        P.solve(model)
        assert P.value(model, x) == pytest.approx(-3.0, abs=1e-8)
        P.solve(model)
        assert P.value(model, x) == pytest.approx(-3.0, abs=1e-8)


def test_issue_29_constant_in_objective():                                 # test/model.jl:198-206
    model = P.Model(DenseQPOptimizer())
    x = Variable(model)
    P.constraint(model, [x], ">=", [0])
    P.objective(model, P.Minimize, x ** 2 + 1)
    P.solve(model)
    assert P.value(model, x) == pytest.approx(0.0, abs=1e-8) and P.objectivevalue(model) == pytest.approx(1.0, abs=1e-8)


def test_scalar_constraints_2_closed_form():                               # test/model.jl:281-300
    model = P.Model(DenseQPOptimizer(variable_offset=5))
    x, y, z = (Variable(model) for _ in range(3))
    P.objective(model, P.Minimize, x ** 2 + x * y + y ** 2 + y * z + z ** 2)
    P.constraint(model, x + 2 * y + 3 * z >= 4)
    P.constraint(model, x + y >= 1)
    P.solve(model)
    assert P.terminationstatus(model) == "OPTIMAL" and P.primalstatus(model) == "FEASIBLE_POINT"
    assert P.objectivevalue(model) == pytest.approx(13 / 7, abs=1e-6)
    np.testing.assert_allclose(P.value(model, [x, y, z]), [4 / 7, 3 / 7, 6 / 7], atol=1e-6)


def test_scalar_constraints_3_closed_form():                               # test/model.jl:302-321
    model = P.Model(DenseQPOptimizer())
    x, y = Variable(model), Variable(model)
    P.objective(model, P.Minimize, 2 * x ** 2 + y ** 2 + x * y + x + y + 1)
    P.constraint(model, x >= 0)
    P.constraint(model, -1 * y <= 0)
    P.constraint(model, x + y, "==", 1)          # (Python's == on host functions is structural equality, as isequal in the tests)
    P.solve(model)
    assert P.terminationstatus(model) == "OPTIMAL" and P.primalstatus(model) == "FEASIBLE_POINT"
    assert P.objectivevalue(model) == pytest.approx(2.875, abs=1e-6)
    np.testing.assert_allclose(P.value(model, [x, y]), [0.25, 0.75], atol=1e-6)


def test_default_objective_and_vector_equality():                          # test/model.jl:323-338
    model = P.Model(DenseQPOptimizer())
    x = Variable(model)
    P.constraint(model, x, "==", 1)
    P.solve(model)
    assert P.terminationstatus(model) == "OPTIMAL" and P.primalstatus(model) == "FEASIBLE_POINT"
    assert P.objectivevalue(model) == 0.0 and P.value(model, x) == pytest.approx(1.0, abs=1e-6)
    model2 = P.Model(DenseQPOptimizer())
    xs = (Variable(model2), Variable(model2))                              # AbstractVector constraints: any sequence of Variables
    P.constraint(model2, xs, "==", (1.0, 2.0))
    P.solve(model2)
    np.testing.assert_allclose(P.value(model2, list(xs)), [1.0, 2.0], atol=1e-8)


def test_moi_issue_426_vector_constraint_lists_by_set_type():              # test/model.jl:208-220
    """the backend lists a model's VectorAffineFunction constraints by set type: none before, one per set after `[x] >= [0]`,
    `[x] <= [1]`, `[x] == [0.5]` (src/moi_interop.jl:180-193: one typed vector per (function, set) pair)"""
    model = P.Model(DenseQPOptimizer())
    x = Variable(model)
    specs = ("vectoraffinefunction_in_nonnegatives", "vectoraffinefunction_in_nonpositives", "vectoraffinefunction_in_zeros")
    for s in specs:
        assert len(model.constraints.by_spec[s]) == 0
    P.constraint(model, [x], ">=", [0])
    P.constraint(model, [x], "<=", [1])
    P.constraint(model, [x], "==", [0.5])
    for s in specs:
        assert len(model.constraints.by_spec[s]) == 1
    assert len(model.constraints) == 3
    P.solve(model)
    assert P.value(model, x) == pytest.approx(0.5, abs=1e-8)


def test_boolean_basics():                                                 # test/model.jl:222-234 (GLPK there; an enumerating stand-in here)
    from qp_solver import TinyMIPOptimizer
    model = P.Model(TinyMIPOptimizer())
    x = Variable(model)
    P.constraint(model, x, "in", "{0, 1}")
    P.objective(model, P.Maximize, x)
    P.solve(model)
    assert P.terminationstatus(model) == "OPTIMAL" and P.primalstatus(model) == "FEASIBLE_POINT"
    assert P.value(model, x) == pytest.approx(1.0, abs=1e-8)
    with pytest.raises(P.ArgumentError):
        P.constraint(P.Model(TinyMIPOptimizer()), x, "in", "{0, 2}")       # src/model.jl:243


@pytest.mark.parametrize("form", ["scalar", "vector"])
def test_integer_basics(form):                                             # test/model.jl:236-267
    from qp_solver import TinyMIPOptimizer
    model = P.Model(TinyMIPOptimizer(variable_offset=3))
    x = Variable(model)
    P.constraint(model, x, "∈", "ℤ")
    if form == "scalar":
        P.constraint(model, x >= 0.5)
    else:
        P.constraint(model, [x], ">=", [0.5])
    P.objective(model, P.Minimize, x)
    P.solve(model)
    assert P.terminationstatus(model) == "OPTIMAL" and P.primalstatus(model) == "FEASIBLE_POINT"
    assert P.value(model, x) == pytest.approx(1.0, abs=1e-8)


def test_nested_expression_of_a_scalar_parameter():                        # test/lazyexpression.jl:39-49
    model = P.mock_model()
    a, b = 3, 4.0
    cval = [5]
    c = P.Parameter(lambda: cval[0], model)
    expr1 = a + b * c
    expr2 = 4 * expr1
    assert expr2() == 4 * expr1() == 4 * (3 + 4.0 * 5)
    cval[0] = 6
    model.setdirty()
    assert expr2() == 4 * (3 + 4.0 * 6)
    repr(expr1)
