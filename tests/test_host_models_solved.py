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
