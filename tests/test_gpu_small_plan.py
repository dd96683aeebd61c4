"""-m gpu: small plans (csrc/small.hip) — a run of small tape entries replayed as ONE launch of the interpreter kernel.

The reference walks README Example 1's DAG at nanoseconds per hop (src/lazyexpression.jl:50-61, src/model.jl:132-143); the fused replay is
what makes the device path competitive there.  Checked here: (1) through the C ABI alone, the fused replay of config 1's whole update!
(four Parameter callbacks with dynamic seeds, residual, literal objective with the MOI copy under a permuted varmap, constraint block)
against the committed golden fixture and the oracle, bit for bit, and against the unfused replay of the same tape; (2) every supported
node kind, fused against unfused; (3) the bounds — a large node stays a launch of its own; (4) through the host API: Model.initialize
selects the small plan by itself and update! is one launch."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class Plan:
    def __init__(self):
        import gpu_util as g
        g.lib()
        self.g = g
        self.plan = C.c_void_p()
        g.call("pmt_plan_create", 0, g.stream(), C.byref(self.plan))
        self.rec = C.c_void_p(g.lib().pmt_plan_recording_stream(self.plan))

    def __enter__(self):
        self.g.call("pmt_plan_begin_record", self.plan)
        return self

    def __exit__(self, *a):
        self.g.call("pmt_plan_end_record", self.plan)

    def fused(self):
        gr, n, ln = C.c_int(), C.c_int(), C.c_int64()
        self.g.call("pmt_plan_fused", self.plan, C.byref(gr), C.byref(n), C.byref(ln))
        return gr.value, n.value, ln.value

    def update(self):
        self.g.call("pmt_plan_update", self.plan)

    def fusion(self, on):
        self.g.call("pmt_plan_set_fusion", self.plan, 1 if on else 0)

    def close(self):
        torch.cuda.synchronize()
        self.g.call("pmt_plan_destroy", self.plan)


def test_config1_update_is_one_launch_and_matches_golden_and_oracle():
    import gpu_util as g
    from oracle import oracle as O
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_readme_example1.npz"))
    n, r, m = int(gold["n"]), int(gold["r"]), int(gold["m"])
    A, b, Cm, d = g.empty_f64(r * n), g.empty_f64(r), g.empty_f64(m * n), g.empty_f64(m)
    xvar, vm = g.to_dev(np.arange(1, n + 1, dtype=np.int64)), g.to_dev(gold["varmap"])
    lt, cst = g.empty_terms(r * n, g.LT), g.empty_f64(r)
    oq, ol, oc = g.empty_terms(r * n * n, g.QT), g.empty_terms(2 * r * n, g.LT), g.empty_f64(1)
    vt, vc = g.empty_terms(m * n, g.VAT), g.empty_f64(m)
    seeds = [C.c_uint64(s) for s in (1, 2, 3, 4)]                     # the golden fixture's Parameter streams
    p = Plan()
    with p:
        g.call("pmt_fill_uniform_dyn_f64", g.ptr(A), r, n, r, C.byref(seeds[0]), 1.0, p.rec)
        g.call("pmt_fill_uniform_dyn_f64", g.ptr(b), r, 1, r, C.byref(seeds[1]), 1.0, p.rec)
        g.call("pmt_fill_uniform_dyn_f64", g.ptr(Cm), m, n, m, C.byref(seeds[2]), 1.0, p.rec)
        g.call("pmt_fill_uniform_dyn_f64", g.ptr(d), m, 1, m, C.byref(seeds[3]), 2.0, p.rec)
        g.call("pmt_affine_assemble_f64", g.ptr(A), r, r, n, g.ptr(xvar), g.ptr(b), -1, g.ptr(lt), g.ptr(cst), p.rec)
        g.call("pmt_quad_expand_f64", r, g.ptr(lt), n, g.ptr(cst), g.ptr(lt), n, g.ptr(cst), 1, g.ptr(vm), g.ptr(oq), g.ptr(ol), g.ptr(oc), p.rec)
        g.call("pmt_affine_pack_vector_f64", g.ptr(Cm), m, m, n, g.ptr(xvar), g.ptr(d), -1, g.ptr(vm), 0, g.ptr(vt), g.ptr(vc), p.rec)
    assert p.fused() == (1, 7, 1)                                     # one run, seven tape entries, ONE launch per update!
    # barriers only between dependent nodes: [four callbacks] | [residual] | [objective, constraint block]
    assert int(g.lib().pmt_plan_fused_phases(p.plan)) == 3
    assert int(g.lib().pmt_plan_tape_length(p.plan)) == 7

    def outputs():
        return (g.terms_to_host(oq, r * n * n, g.QT), g.terms_to_host(ol, 2 * r * n, g.LT), g.f64_to_host(oc, 1),
                g.terms_to_host(vt, m * n, g.VAT), g.f64_to_host(vc, m), g.f64_to_host(A, r * n))

    p.update()
    q, l, c, v, vcs, a_host = outputs()
    assert g.same_bits(a_host, gold["A"])
    g.assert_terms_equal(q, gold["literal_quad"])
    g.assert_terms_equal(l, gold["literal_aff"])
    assert g.same_bits(c, gold["const"])
    g.assert_terms_equal(v, gold["constraint_terms"])
    assert g.same_bits(vcs, gold["constraint_consts"])
    # the next update!: every callback draws its next stream (seed + 1000 per epoch); against the oracle, and against the unfused replay
    for epoch in (1, 2):
        for k, w in enumerate(seeds):
            w.value = k + 1 + 1000 * epoch
        p.update()
        fused = outputs()
        Ah = O.fill_uniform(r * n, 1 + 1000 * epoch); bh = O.fill_uniform(r, 2 + 1000 * epoch)
        Ch = O.fill_uniform(m * n, 3 + 1000 * epoch); dh = O.fill_uniform(m, 4 + 1000 * epoch, 2.0)
        w_ = O.LsqWorkspace(n, r, m)
        xi = np.arange(1, n + 1, dtype=np.int64)
        w_.eval_objective(Ah, bh, xi); w_.eval_constraint(Ch, dh, xi)
        at, qt, const = w_.objective.moi(gold["varmap"])
        ct, cc = w_.constraint.moi(gold["varmap"])
        assert np.array_equal(fused[0].view(np.int64), qt.view(np.int64)) and np.array_equal(fused[1].view(np.int64), at.view(np.int64))
        assert fused[2][0] == const and np.array_equal(fused[3].view(np.int64), ct.view(np.int64)) and np.array_equal(fused[4], cc)
        for t in (oq, ol, vt):
            t.fill_(-7)
        p.fusion(False)
        assert p.fused() == (0, 0, 7)
        p.update()
        plain = outputs()
        p.fusion(True)
        for f_, u_ in zip(fused, plain):
            assert np.array_equal(np.ascontiguousarray(f_).view(np.int64), np.ascontiguousarray(u_).view(np.int64))
    p.close()


def test_every_node_kind_fused_equals_unfused():
    """vars_addsub (bounds), consts, the three MOI packs of materialised functions (uniform and ragged rows), copy, static-seed fills, the
    native (moi = 0) literal objective: one tape, replayed fused and as recorded; identical bytes in every output buffer"""
    import gpu_util as g
    rng = np.random.default_rng(5)
    n, rows = 11, 5
    vmh = rng.permutation(n).astype(np.int64) + 1
    xvar, vm = g.to_dev(np.arange(1, n + 1, dtype=np.int64)), g.to_dev(vmh)
    A, b, lows = g.empty_f64(rows * n), g.empty_f64(rows), g.empty_f64(n)
    lt, cst = g.empty_terms(rows * n, g.LT), g.empty_f64(rows)
    q0, l0, c0 = g.empty_terms(rows * n * n, g.QT), g.empty_terms(2 * rows * n, g.LT), g.empty_f64(1)
    sq, sa = g.empty_terms(rows * n * n, g.QT), g.empty_terms(2 * rows * n, g.LT)
    va = g.empty_terms(rows * n, g.VAT)
    ragged_ptr = np.array([0, 3, 3, 10, 11, rows * n], dtype=np.int64)
    va2, rp = g.empty_terms(rows * n, g.VAT), g.to_dev(ragged_ptr)
    bt_lt, bt_vat, bc = g.empty_terms(n, g.LT), g.empty_terms(n, g.VAT), g.empty_f64(n)
    cs, cp = g.empty_f64(rows), g.empty_f64(rows)
    outs = [lt, cst, q0, l0, c0, sq, sa, va, va2, bt_lt, bt_vat, bc, cs, cp]
    p = Plan()
    with p:
        g.call("pmt_fill_uniform_matrix_f64", g.ptr(A), rows, n, rows, C.c_uint64(11), 1.0, p.rec)
        g.call("pmt_fill_uniform_f64", g.ptr(b), rows, C.c_uint64(12), 1.0, p.rec)
        g.call("pmt_fill_uniform_f64", g.ptr(lows), n, C.c_uint64(13), -1.0, p.rec)
        g.call("pmt_affine_assemble_f64", g.ptr(A), rows, rows, n, g.ptr(xvar), g.ptr(b), 1, g.ptr(lt), g.ptr(cst), p.rec)
        g.call("pmt_quad_expand_f64", rows, g.ptr(lt), n, g.ptr(cst), g.ptr(lt), n, g.ptr(cst), 0, None, g.ptr(q0), g.ptr(l0), g.ptr(c0), p.rec)
        g.call("pmt_pack_scalar_quadratic_f64", g.ptr(q0), rows * n * n, g.ptr(vm), g.ptr(sq), p.rec)
        g.call("pmt_pack_scalar_affine_f64", g.ptr(l0), 2 * rows * n, g.ptr(vm), g.ptr(sa), p.rec)
        g.call("pmt_pack_vector_affine_f64", g.ptr(lt), None, rows, n, g.ptr(vm), 3, g.ptr(va), p.rec)
        g.call("pmt_pack_vector_affine_f64", g.ptr(lt), g.ptr(rp), rows, 0, g.ptr(vm), 0, g.ptr(va2), p.rec)
        g.call("pmt_vars_addsub_f64", g.ptr(xvar), n, g.ptr(lows), -1, g.ptr(vm), 2, g.ptr(bt_lt), g.ptr(bt_vat), g.ptr(bc), p.rec)
        g.call("pmt_consts_f64", g.ptr(b), rows, -1, g.ptr(cs), p.rec)
        g.call("pmt_copy_bytes", g.ptr(cp), g.ptr(cs), 8 * rows, p.rec)
    groups, nodes, ln = p.fused()
    # the ragged pack's size lives on the device: it stays a launch of its own and cuts the run in two
    assert (groups, nodes, ln) == (2, 11, 3)
    p.update()
    torch.cuda.synchronize()
    fused = [t.clone() for t in outs]
    for t in outs:
        t.fill_(-7) if t.dtype == torch.int64 else t.fill_(float("nan"))
    p.fusion(False)
    p.update()
    torch.cuda.synchronize()
    for f_, t in zip(fused, outs):
        assert torch.equal(f_.view(torch.int64), t.view(torch.int64))
    # and the values mean what they should: the bounds rows, the doubled diagonal of the scalar quadratic pack
    sqh, q0h = g.terms_to_host(sq, rows * n * n, g.QT), g.terms_to_host(q0, rows * n * n, g.QT)
    diag = q0h["row"] == q0h["col"]
    assert np.array_equal(sqh["coeff"][diag], 2 * q0h["coeff"][diag]) and np.array_equal(sqh["coeff"][~diag], q0h["coeff"][~diag])
    assert np.array_equal(sqh["row"], vmh[q0h["row"] - 1]) and np.array_equal(sqh["col"], vmh[q0h["col"] - 1])
    bth = g.terms_to_host(bt_vat, n, g.VAT)
    assert np.array_equal(bth["out"], np.arange(3, n + 3)) and np.all(bth["coeff"] == 1.0) and np.array_equal(bth["var"], vmh)
    p.close()


def test_algebra_node_kinds_fused_equals_unfused():
    """the builders behind the other rewrite rules — adjoint, vcat!/add!/subtract! (uniform rows), scale!/mul! (host and device scalar),
    matvecmul! over affine functions, the vecdot! forms, bilinearmul! — as small-plan nodes: ONE launch for the whole tape, identical
    bytes to the tape replayed as recorded"""
    import gpu_util as g
    rng = np.random.default_rng(9)
    n, rows, L = 6, 4, 6
    vmh = rng.permutation(n + 3).astype(np.int64) + 1
    xvar, yvar, vm = g.to_dev(np.arange(1, n + 1, dtype=np.int64)), g.to_dev(np.array([3, 1, 6, 2, 9, 4], dtype=np.int64)), g.to_dev(vmh)
    A, At, b, w = g.empty_f64(rows * n), g.empty_f64(rows * n), g.empty_f64(rows), g.empty_f64(n)
    sdev = g.to_dev(np.array([1.75]))
    lt, cst = g.empty_terms(rows * n, g.LT), g.empty_f64(rows)               # A x + b
    lt2, cst2 = g.empty_terms(n * rows, g.LT), g.empty_f64(n)                 # A' y4 (transpose feeding assemble)
    comb, combc = g.empty_terms(rows * 2 * n, g.LT), g.empty_f64(rows)        # [r ; -r]
    sc, scc = g.empty_terms(rows * n, g.LT), g.empty_f64(rows)                # 1.75 * r   (device scalar)
    sc2, scc2 = g.empty_terms(rows * n, g.LT), g.empty_f64(rows)              # 0.5 * r    (host scalar)
    B = g.empty_f64(3 * rows)
    mv, mvc = g.empty_terms(3 * rows * n, g.LT), g.empty_f64(3)               # B (A x + b)
    nv, nvc = g.empty_terms(n, g.LT), g.empty_f64(1)                          # w . x
    na, nac = g.empty_terms(rows * n, g.LT), g.empty_f64(1)                   # b . r
    q1 = g.empty_terms(n, g.QT)                                               # (w x) . y terms
    q2, l2 = g.empty_terms(rows * n, g.QT), g.empty_terms(rows, g.LT)         # r . x[:rows]
    Qm = g.empty_f64(n * n)
    q3 = g.empty_terms(n * n, g.QT)                                           # x' Q y
    qc, qs = g.empty_terms(n + rows * n, g.QT), g.empty_terms(n + rows * n, g.QT)
    sv, sn = g.empty_terms(n, g.LT), g.empty_f64(n)
    outs = [At, lt, cst, lt2, cst2, comb, combc, sc, scc, sc2, scc2, mv, mvc, nv, nvc, na, nac, q1, q2, l2, q3, qc, qs, sv, sn]
    p = Plan()
    with p:
        g.call("pmt_fill_uniform_matrix_f64", g.ptr(A), rows, n, rows, C.c_uint64(31), 1.0, p.rec)
        g.call("pmt_fill_uniform_f64", g.ptr(b), rows, C.c_uint64(32), 1.0, p.rec)
        g.call("pmt_fill_uniform_f64", g.ptr(w), n, C.c_uint64(33), 1.0, p.rec)
        g.call("pmt_fill_uniform_f64", g.ptr(B), 3 * rows, C.c_uint64(34), 1.0, p.rec)
        g.call("pmt_fill_uniform_f64", g.ptr(Qm), n * n, C.c_uint64(35), 1.0, p.rec)
        g.call("pmt_transpose_f64", g.ptr(A), rows, rows, n, g.ptr(At), n, p.rec)
        g.call("pmt_affine_assemble_f64", g.ptr(A), rows, rows, n, g.ptr(xvar), g.ptr(b), 1, g.ptr(lt), g.ptr(cst), p.rec)
        g.call("pmt_affine_assemble_f64", g.ptr(At), n, n, rows, g.ptr(yvar), None, 0, g.ptr(lt2), g.ptr(cst2), p.rec)
        g.call("pmt_affvec_combine_f64", rows, g.ptr(lt), None, n, g.ptr(cst), g.ptr(lt), None, n, g.ptr(cst), -1, g.ptr(comb), None, 2 * n, g.ptr(combc), p.rec)
        g.call("pmt_affvec_scale_f64", rows, rows * n, g.ptr(lt), g.ptr(cst), g.ptr(sdev), 0.0, g.ptr(sc), g.ptr(scc), p.rec)
        g.call("pmt_affvec_scale_f64", rows, rows * n, g.ptr(lt), g.ptr(cst), None, 0.5, g.ptr(sc2), g.ptr(scc2), p.rec)
        g.call("pmt_matvecmul_affs_f64", g.ptr(B), 3, 3, rows, g.ptr(lt), n, g.ptr(cst), g.ptr(mv), g.ptr(mvc), p.rec)
        g.call("pmt_vecdot_numbers_vars_f64", g.ptr(w), g.ptr(xvar), n, g.ptr(nv), g.ptr(nvc), p.rec)
        g.call("pmt_vecdot_numbers_affs_f64", g.ptr(b), rows, g.ptr(lt), n, g.ptr(cst), g.ptr(na), g.ptr(nac), p.rec)
        g.call("pmt_vecdot_terms_f64", n, g.ptr(w), g.ptr(xvar), None, g.ptr(yvar), 1, g.ptr(vm), g.ptr(q1), p.rec)
        g.call("pmt_vecdot_affs_vars_f64", rows, g.ptr(lt), n, g.ptr(cst), g.ptr(xvar), 1, g.ptr(vm), g.ptr(q2), g.ptr(l2), p.rec)
        g.call("pmt_bilinear_f64", g.ptr(Qm), n, n, n, g.ptr(xvar), g.ptr(yvar), 1, g.ptr(vm), g.ptr(q3), p.rec)
        g.call("pmt_quad_combine_f64", g.ptr(q1), n, g.ptr(q2), rows * n, -1, g.ptr(qc), p.rec)
        g.call("pmt_quad_scale_f64", g.ptr(qc), n + rows * n, g.ptr(sdev), 0.0, g.ptr(qs), p.rec)
        g.call("pmt_scale_vars_f64", g.ptr(yvar), n, None, 2.5, g.ptr(sv), p.rec)
        g.call("pmt_scale_numbers_f64", g.ptr(w), n, g.ptr(sdev), 0.0, g.ptr(sn), p.rec)
    assert p.fused() == (1, 21, 1)
    assert int(g.lib().pmt_plan_fused_phases(p.plan)) < 21
    p.update()
    torch.cuda.synchronize()
    fused = [t.clone() for t in outs]
    for t in outs:
        t.fill_(-7) if t.dtype == torch.int64 else t.fill_(float("nan"))
    p.fusion(False)
    assert p.fused() == (0, 0, 21)
    p.update()
    torch.cuda.synchronize()
    for k, (f_, t) in enumerate(zip(fused, outs)):
        assert torch.equal(f_.view(torch.int64), t.view(torch.int64)), k
    # a few values against their meaning (the oracle-level checks of these builders are tests/test_gpu_kernels.py and test_gpu_rules_vs_oracle.py)
    Ah = g.f64_to_host(A, rows * n).reshape(n, rows).T
    assert np.array_equal(g.f64_to_host(At, rows * n).reshape(rows, n), Ah)               # At is n x rows column-major = A row-major
    q3h = g.terms_to_host(q3, n * n, g.QT)
    Qh = g.f64_to_host(Qm, n * n)
    xv, yv = np.arange(1, n + 1), np.array([3, 1, 6, 2, 9, 4])
    want = np.array([Qh[e] * (2.0 if xv[e // n] == yv[e % n] else 1.0) for e in range(n * n)])   # Q[k] in column-major LINEAR order (functions.jl:853)
    assert np.array_equal(q3h["coeff"], want) and np.array_equal(q3h["row"], vmh[xv[np.arange(n * n) // n] - 1])
    assert np.array_equal(g.f64_to_host(sn, n), 1.75 * g.f64_to_host(w, n))
    p.close()


def test_large_nodes_keep_their_own_kernels():
    """the interpreter is one workgroup: an entry that writes more than 32768 elements is replayed by its own kernel, and a run is cut
    before it exceeds 65536"""
    import gpu_util as g
    n, r = 256, 256                                               # 65536 LinearTerms: above the per-node bound
    A, b = g.empty_f64(r * n), g.empty_f64(r)
    xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    lt, cst = g.empty_terms(r * n, g.LT), g.empty_f64(r)
    small_out = [g.empty_f64(20000) for _ in range(5)]
    p = Plan()
    with p:
        g.call("pmt_fill_uniform_f64", g.ptr(b), r, C.c_uint64(2), 1.0, p.rec)
        g.call("pmt_fill_uniform_matrix_f64", g.ptr(A), r, n, r, C.c_uint64(1), 1.0, p.rec)                   # 65536 elements: alone
        g.call("pmt_affine_assemble_f64", g.ptr(A), r, r, n, g.ptr(xvar), g.ptr(b), -1, g.ptr(lt), g.ptr(cst), p.rec)   # alone
        for k, t in enumerate(small_out):                                                                      # 5 x 20000: runs of 3 + 2
            g.call("pmt_fill_uniform_f64", g.ptr(t), 20000, C.c_uint64(20 + k), 1.0, p.rec)
    assert p.fused() == (2, 5, 5)          # [b] [A] [assemble] [3 fills] [2 fills]
    p.update()
    from oracle import oracle as O
    for k, t in enumerate(small_out):
        assert g.same_bits(g.f64_to_host(t, 20000), O.fill_uniform(20000, 20 + k))
    got = g.terms_to_host(lt, r * n, g.LT)
    assert g.same_bits(got["coeff"].reshape(r, n), O.fill_uniform(r * n, 1).reshape(n, r).T)
    p.close()


@pytest.mark.parametrize("mode", ["literal", "canonical"])
def test_model_selects_the_small_plan_by_itself(mode):
    """README Example 1 through the host API with device-side callbacks: Model.initialize records the callbacks into the tape and the
    library fuses; update! is ONE launch (literal) and what the optimizer receives equals the oracle for every epoch's Parameter values"""
    import parametron_jl_amd as P
    from oracle import oracle as O
    from qp_solver import DenseQPOptimizer
    n, m = 8, 2
    opt = DenseQPOptimizer(variable_offset=0, permute_seed=7)
    model = P.Model(opt, quadratic_mode=mode)
    x = [P.Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((n, n), 1, model); b = P.DeviceUniformParameter((n,), 2, model)
    Cm = P.DeviceUniformParameter((m, n), 3, model); d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    P.constraint(model, Cm * x == d)
    for it in range(4):
        P.solve(model)
        fz = model.device().fused()
        # every kernel of the update! — four callbacks, residual, objective, constraint — is ONE launch (literal: 10 tape entries — the MOI copy of a materialised objective is two packs and the copy of its constant into the function object's page-locked word; canonical:
        # the tiny Gram node is a small-plan node too, 6 entries); a small model fetches its MOI buffers with plain copies behind the replay
        assert fz["groups"] == 1 and fz["exec_length"] == 1 and fz["nodes"] == model.device().tape_length() == (10 if mode == "literal" else 6), fz
        P.profile_enable(True)
        model.setdirty(); model._run_tape(fetch=False); model.device().synchronize()
        rep = P.profile_report()
        P.profile_enable(False)
        assert list(rep) == ["small_plan_kernel"] and rep["small_plan_kernel"]["launches"] == 1, rep
        P.solve(model)
        assert all(getattr(p_, "_in_tape", False) for p_ in (A, b, Cm, d))
        vm = model.model_var_to_optimizer
        w = O.LsqWorkspace(n, n, m)
        xi = np.arange(1, n + 1, dtype=np.int64)
        Ah, bh, Ch, dh = (O.fill_uniform(int(np.prod(p_.shape)), p_.current_seed(), p_.scale) for p_ in (A, b, Cm, d))
        assert g_same(np.asfortranarray(A()).reshape(-1, order="F"), Ah) and g_same(d(), dh)       # the host copies are this epoch's values
        w.eval_objective(Ah, bh, xi); w.eval_constraint(Ch, dh, xi)
        f = model.objective.f
        if mode == "canonical":
            w.objective.canonicalize()
        at, qt, const = w.objective.moi(vm)
        assert np.array_equal(f.quadratic_terms["row"], qt["row"]) and np.array_equal(f.quadratic_terms["col"], qt["col"])
        if mode == "literal":
            assert np.array_equal(f.quadratic_terms.view(np.int64), qt.view(np.int64)) and np.array_equal(f.affine_terms.view(np.int64), at.view(np.int64))
        else:
            np.testing.assert_allclose(f.quadratic_terms["coeff"], qt["coeff"], rtol=1e-12, atol=0)
            np.testing.assert_allclose(f.affine_terms["coeff"], at["coeff"], rtol=1e-12, atol=0)
        assert f.constant == const
        ct, cc = w.constraint.moi(vm)
        cf = list(model.constraints)[0].f
        assert np.array_equal(cf.terms.view(np.int64), ct.view(np.int64)) and np.array_equal(cf.constants, cc)
    model.close()


def g_same(a, b):
    import gpu_util as g
    return g.same_bits(a, b)


def test_host_callbacks_of_a_small_model_travel_through_mailboxes():
    """README Example 1 as the reference writes it — HOST callbacks `rand!(A)` (src/parameter.jl:57, README.md:36-43): the values of a small
    model's host-updated Parameters reach the device through page-locked mailboxes that the first entries of the tape copy from, inside the
    one small-plan launch; update! issues no per-Parameter upload.  What the optimizer receives equals the oracle for every solve, also
    when a value is read or changed outside update! in between."""
    import parametron_jl_amd as P
    from oracle import oracle as O
    from qp_solver import DenseQPOptimizer
    n, m = 8, 2
    model = P.Model(DenseQPOptimizer(variable_offset=2, permute_seed=3), quadratic_mode="literal")
    x = [P.Variable(model) for _ in range(n)]
    rng = np.random.default_rng(4)

    def randrng(a):
        a[...] = rng.random(a.shape)
    A = P.Parameter(randrng, np.zeros((n, n)), model)
    b = P.Parameter(randrng, np.zeros(n), model)
    Cm = P.Parameter(model, val=rng.random((m, n)))               # a manually updated work buffer (src/parameter.jl:88)
    d = P.Parameter(randrng, np.zeros(m), model)
    s = P.Parameter(lambda: 1.5, model)                            # an out-of-place scalar callback
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    P.constraint(model, s * (Cm * x) == d)

    def check():
        vm = model.model_var_to_optimizer
        w = O.LsqWorkspace(n, n, m)
        xi = np.arange(1, n + 1, dtype=np.int64)
        w.eval_objective(np.asfortranarray(A()).reshape(-1, order="F"), b(), xi)
        at, qt, const = w.objective.moi(vm)
        f = model.objective.f
        assert np.array_equal(f.quadratic_terms.view(np.int64), qt.view(np.int64)) and np.array_equal(f.affine_terms.view(np.int64), at.view(np.int64))
        assert f.constant == const
        cf = list(model.constraints)[0].f
        ref = O.AffVec(m).vecsubtract(O.AffVec(m).scale_number_affs(1.5, O.AffVec(m).matvecmul_vars(Cm(), list(xi))), d())
        ct, cc = ref.moi(vm)
        assert np.array_equal(cf.terms.view(np.int64), ct.view(np.int64)) and np.array_equal(cf.constants, cc)

    for it in range(5):
        if it == 2:
            Cm.val[...] = rng.random((m, n))                       # the user rewrites the buffer between solves
        if it == 3:
            model.setdirty(); res()                                # a lazy expression evaluated OUTSIDE update!: same values afterwards
        P.solve(model)
        check()
        fz = model.device().fused()
        assert fz["groups"] == 1 and fz["exec_length"] == 1, fz
        for p_ in (A, b, Cm, d, s):
            assert getattr(p_, "_mailbox", None) is not None and p_._in_tape
    # the mailbox holds the device layout of the current value
    assert np.array_equal(A._mailbox.reshape(n, A._dev.lda)[:, :n].T, A()) and s._mailbox[0] == 1.5
    P.profile_enable(True)
    model.setdirty(); model._run_tape(fetch=False); model.device().synchronize()
    rep = P.profile_report()
    P.profile_enable(False)
    assert list(rep) == ["small_plan_kernel"], rep
    model.close()


def test_small_model_with_several_constraints_follows_every_update():
    """A small model whose constraints are built straight from host-updated Parameters (the records a large model puts on the plan's side
    lane): its Parameter values arrive through mailbox copies at the FRONT of the tape, so every record stays on the plan's own lane, in the
    one-launch run, behind those copies — and what the function objects hold after each solve! (stored by the kernels into their page-locked
    arrays: no device twin, no D2H copy) is THIS solve's data, checked against numpy on the host buffers.  (With the constraints on the
    side lane they forked at the top of the replay, in front of the copies: stale by one solve whenever the objective's node was fast.)"""
    import parametron_jl_amd as P
    rng = np.random.default_rng(9)
    n, r, m = 60, 90, 12
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical")
    x = [P.Variable(model) for _ in range(n)]
    bufs = {"A": np.zeros((r, n), order="F"), "b": np.zeros(r), "G": np.zeros((m, n), order="F"), "h": np.zeros(m), "l": np.zeros(n)}
    A, b, G, h, lo = (P.Parameter(model, val=bufs[k]) for k in ("A", "b", "G", "h", "l"))
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    cG = P.constraint(model, G * x, "<=", h)
    cl = P.constraint(model, x, ">=", lo)
    P.solve(model)
    assert model._small and not getattr(model, "_lane_records", [])
    fz = model.device().fused()
    assert fz["groups"] == 1 and fz["exec_length"] == 2, fz            # the run of small entries + the objective's node (emitted last)
    recs = sorted((c for c in model.constraints), key=lambda c: len(c.f.constants))
    assert [len(c.f.constants) for c in recs] == [m, n]
    for rec in list(recs) + [model.objective]:                           # no device twin: the "device" address of every MOI buffer is the host array
        for key, host in (("terms", getattr(rec.f, "_terms", None)), ("consts", getattr(rec.f, "constants", None)),
                          ("quad", getattr(rec.f, "quadratic_terms", None)), ("lin", getattr(rec.f, "affine_terms", None))):
            if host is not None and key in rec.dev:
                assert rec.dev[key] == host.ctypes.data, (type(rec.f).__name__, key)
    for it in range(25):
        for v in bufs.values():
            v[...] = rng.random(v.shape)
        P.solve(model)
        f = model.objective.f
        iu = np.triu_indices(n)
        np.testing.assert_allclose(f.quadratic_terms["coeff"], (2 * bufs["A"].T @ bufs["A"])[iu], rtol=1e-12, atol=0)
        np.testing.assert_allclose(f.affine_terms["coeff"], -2 * bufs["A"].T @ bufs["b"], rtol=1e-12, atol=0)
        assert f.constant == pytest.approx(bufs["b"] @ bufs["b"], rel=1e-13)
        assert np.array_equal(recs[0].f.constants, 0.0 - bufs["h"]) and np.array_equal(recs[1].f.constants, 0.0 - bufs["l"])
        t = recs[0].f.terms                                              # VectorAffineTerms of G x: row-major (src/moi_interop.jl:64-81)
        assert np.array_equal(t["coeff"].reshape(m, n), bufs["G"]) and np.all(recs[1].f.terms["coeff"] == 1.0)
    model.close()


def test_runs_on_several_workgroups_equal_the_recorded_tape():
    """A run with tens of thousands of elements of work is executed by SEVERAL workgroups; the barrier in front of its dependent nodes is a
    grid barrier on a counter the plan owns (csrc/small.hip), whose base advances with every replay.  Every replay: the constraint's MOI
    terms and constants equal the oracle bit for bit for this replay's seeds (phase 2 reads, across workgroups, what phase 1's fills
    wrote), the objective the tall node's tolerance; the same tape replayed as recorded gives the same bits; a hipGraph of the plan falls
    back to single-workgroup runs."""
    import parametron_jl_amd as P
    from oracle import oracle as O
    n, r, m = 100, 150, 30
    model = P.Model(P.MockOptimizer(), quadratic_mode="canonical")
    x = [P.Variable(model) for _ in range(n)]
    A = P.DeviceUniformParameter((r, n), 1, model); b = P.DeviceUniformParameter((r,), 2, model)
    Cm = P.DeviceUniformParameter((m, n), 3, model); d = P.DeviceUniformParameter((m,), 4, model, scale=2.0)
    res = A * x - b
    P.objective(model, P.Minimize, P.dot(res, res))
    P.constraint(model, Cm * x == d)
    P.solve(model)
    ctx = model.device()
    fz = ctx.fused()
    assert model._small and fz["groups"] == 1 and fz["workgroups"] > 1 and fz["phases"] >= 2, fz
    xi = np.arange(1, n + 1, dtype=np.int64)
    vm = model.model_var_to_optimizer
    for it in range(40):
        P.solve(model)
        Ch, dh = (O.fill_uniform(int(np.prod(p_.shape)), p_.current_seed(), p_.scale) for p_ in (Cm, d))
        w = O.LsqWorkspace(n, r, m)
        w.eval_constraint(Ch, dh, xi)
        ct, cc = w.constraint.moi(vm)
        cf = list(model.constraints)[0].f
        assert np.array_equal(cf.terms.view(np.int64), ct.view(np.int64)) and np.array_equal(cf.constants, cc), it
        Ah, bh = (O.fill_uniform(int(np.prod(p_.shape)), p_.current_seed(), p_.scale) for p_ in (A, b))
        Am = Ah.reshape(n, r).T
        f = model.objective.f
        np.testing.assert_allclose(f.quadratic_terms["coeff"], (2 * Am.T @ Am)[np.triu_indices(n)], rtol=1e-12, atol=0)
        np.testing.assert_allclose(f.affine_terms["coeff"], -2 * Am.T @ bh, rtol=1e-12, atol=0)
    fused = [a.copy() for a in (model.objective.f.quadratic_terms, model.objective.f.affine_terms, cf.terms, cf.constants)]
    ctx.synchronize(); ctx.set_fusion(False)
    assert ctx.fused()["groups"] == 0
    model._run_tape(); ctx.synchronize()                                  # the same Parameter values (nothing was set dirty), entry by entry
    for a, b_ in zip(fused, (model.objective.f.quadratic_terms, model.objective.f.affine_terms, cf.terms, cf.constants)):
        assert a.tobytes() == b_.tobytes()
    ctx.set_fusion(True)
    assert ctx.fused()["workgroups"] > 1
    model.close()


def test_a_graph_replays_large_runs_with_one_workgroup():
    """C ABI: a recorded run of ~45000 elements takes several workgroups when replayed as launches; pmt_plan_instantiate_graph rebuilds it as
    a single-workgroup run (the grid barrier's base is a kernel argument a captured launch could not advance) — same bits either way"""
    import gpu_util as g
    from oracle import oracle as O
    r, n = 150, 100
    s = g.stream()
    dA, db = g.empty_f64(r * n), g.empty_f64(r)
    xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    res, rc = g.empty_terms(r * n, g.LT), g.empty_f64(r)
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, s, C.byref(plan))
    rec = C.c_void_p(g.lib().pmt_plan_recording_stream(plan))
    g.call("pmt_plan_begin_record", plan)
    g.call("pmt_fill_uniform_matrix_f64", g.ptr(dA), r, n, r, C.c_uint64(5), 1.0, rec)
    g.call("pmt_fill_uniform_f64", g.ptr(db), r, C.c_uint64(6), 1.0, rec)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), r, r, n, g.ptr(xvar), g.ptr(db), -1, g.ptr(res), g.ptr(rc), rec)
    g.call("pmt_plan_end_record", plan)
    assert g.lib().pmt_plan_fused_workgroups(plan) > 1
    want_t = O.AffVec(r).vecsubtract(O.AffVec(r).matvecmul_vars(O.fill_uniform(r * n, 5).reshape(n, r).T.copy(), np.arange(1, n + 1, dtype=np.int64)),
                                     O.fill_uniform(r, 6))
    terms, _, consts = want_t.flat()
    for _ in range(5):                                                    # launches: the counter's base advances
        res.zero_(); rc.zero_()
        g.call("pmt_plan_update", plan)
        torch.cuda.synchronize()
        g.assert_terms_equal(g.terms_to_host(res, r * n, g.LT), terms)
        assert g.same_bits(g.f64_to_host(rc, r), consts)
    g.call("pmt_plan_instantiate_graph", plan)
    assert g.lib().pmt_plan_fused_workgroups(plan) == 1
    for _ in range(3):
        res.zero_(); rc.zero_()
        g.call("pmt_plan_update", plan)
        torch.cuda.synchronize()
        g.assert_terms_equal(g.terms_to_host(res, r * n, g.LT), terms)
        assert g.same_bits(g.f64_to_host(rc, r), consts)
    g.call("pmt_plan_destroy", plan)


def test_grid_barrier_timeout_is_reported_by_synchronize():
    """fault injection 2: every workgroup of a multi-workgroup run waits at its grid barrier for an arrival that never comes (bound cut to
    20 ms); the run finishes, pmt_plan_synchronize returns PMT_HIP_ERROR for that re-evaluation, and the next one (injection off) is clean"""
    import gpu_util as g
    from parametron_jl_amd import _lib
    r, n = 150, 100
    s = g.stream()
    dA, db = g.empty_f64(r * n), g.empty_f64(r)
    xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    res, rc = g.empty_terms(r * n, g.LT), g.empty_f64(r)
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, s, C.byref(plan))
    rec = C.c_void_p(g.lib().pmt_plan_recording_stream(plan))
    g.call("pmt_plan_begin_record", plan)
    g.call("pmt_fill_uniform_matrix_f64", g.ptr(dA), r, n, r, C.c_uint64(5), 1.0, rec)
    g.call("pmt_fill_uniform_f64", g.ptr(db), r, C.c_uint64(6), 1.0, rec)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), r, r, n, g.ptr(xvar), g.ptr(db), -1, g.ptr(res), g.ptr(rc), rec)
    g.call("pmt_plan_end_record", plan)
    assert g.lib().pmt_plan_fused_workgroups(plan) > 1
    g.call("pmt_plan_update", plan); g.call("pmt_plan_synchronize", plan)
    good = g.terms_to_host(res, r * n, g.LT).copy()
    try:
        g.call("pmt_set_fault_injection", 2)
        g.call("pmt_plan_update", plan)
        with pytest.raises(_lib.HipError, match="grid barrier"):
            g.call("pmt_plan_synchronize", plan)
    finally:
        g.call("pmt_set_fault_injection", 0)
    res.zero_()
    g.call("pmt_plan_update", plan); g.call("pmt_plan_synchronize", plan)
    g.assert_terms_equal(g.terms_to_host(res, r * n, g.LT), good)
    g.call("pmt_plan_destroy", plan)


def _large_run_plan(g, stream=None, seed=5):
    """a recorded run of ~45000 elements (fills + affine assemble): several workgroups when replayed as launches"""
    r, n = 150, 100
    s = stream if stream is not None else g.stream()
    dA, db = g.empty_f64(r * n), g.empty_f64(r)
    xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    res, rc = g.empty_terms(r * n, g.LT), g.empty_f64(r)
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, s, C.byref(plan))
    rec = C.c_void_p(g.lib().pmt_plan_recording_stream(plan))
    g.call("pmt_plan_begin_record", plan)
    g.call("pmt_fill_uniform_matrix_f64", g.ptr(dA), r, n, r, C.c_uint64(seed), 1.0, rec)
    g.call("pmt_fill_uniform_f64", g.ptr(db), r, C.c_uint64(seed + 1), 1.0, rec)
    g.call("pmt_affine_assemble_f64", g.ptr(dA), r, r, n, g.ptr(xvar), g.ptr(db), -1, g.ptr(res), g.ptr(rc), rec)
    g.call("pmt_plan_end_record", plan)
    return plan, (dA, db, xvar, res, rc), (r, n)


def test_a_multi_workgroup_run_launched_on_one_workgroup_gives_the_same_bits():
    """while another plan's multi-workgroup run is in flight on the device a run goes out on ONE workgroup (fault injection 4
    forces that path).  Launches on several / one / several workgroups in turn: the same bits every time, and the grid barrier's counter stays
    consistent (a single-workgroup launch never arrives at it)"""
    import gpu_util as g
    plan, (dA, db, xvar, res, rc), (r, n) = _large_run_plan(g)
    assert g.lib().pmt_plan_fused_workgroups(plan) > 1
    g.call("pmt_plan_update", plan); g.call("pmt_plan_synchronize", plan)
    good, goodc = g.terms_to_host(res, r * n, g.LT).copy(), g.f64_to_host(rc, r).copy()
    try:
        for inject in (4, 0, 4, 4, 0, 0):
            g.call("pmt_set_fault_injection", inject)
            res.zero_(); rc.zero_()
            g.call("pmt_plan_update", plan); g.call("pmt_plan_synchronize", plan)
            g.assert_terms_equal(g.terms_to_host(res, r * n, g.LT), good)
            assert g.same_bits(g.f64_to_host(rc, r), goodc)
    finally:
        g.call("pmt_set_fault_injection", 0)
    g.call("pmt_plan_destroy", plan)


def test_two_plans_with_multi_workgroup_runs_on_two_streams():
    """two plans replaying fused runs on several workgroups from two streams, interleaved without synchronisation in between: at most one
    multi-workgroup run is in flight per device (the other launch goes out on one workgroup) — every result right, no barrier time-out"""
    import gpu_util as g
    s2 = torch.cuda.Stream()
    plan1, bufs1, (r, n) = _large_run_plan(g, seed=5)
    with torch.cuda.stream(s2):
        plan2, bufs2, _ = _large_run_plan(g, stream=C.c_void_p(s2.cuda_stream), seed=9)
    torch.cuda.synchronize()
    want = {}
    for name, plan, bufs in (("1", plan1, bufs1), ("2", plan2, bufs2)):
        g.call("pmt_plan_update", plan); g.call("pmt_plan_synchronize", plan)
        want[name] = (g.terms_to_host(bufs[3], r * n, g.LT).copy(), g.f64_to_host(bufs[4], r).copy())
    assert not np.array_equal(want["1"][0]["coeff"], want["2"][0]["coeff"])
    for _ in range(20):
        for _ in range(5):
            g.call("pmt_plan_update", plan1); g.call("pmt_plan_update", plan2)
        for name, plan, bufs in (("1", plan1, bufs1), ("2", plan2, bufs2)):
            g.call("pmt_plan_synchronize", plan)
            g.assert_terms_equal(g.terms_to_host(bufs[3], r * n, g.LT), want[name][0])
            assert g.same_bits(g.f64_to_host(bufs[4], r), want[name][1])
    g.call("pmt_plan_destroy", plan1); g.call("pmt_plan_destroy", plan2)


def test_plan_check_reports_a_barrier_timeout_behind_a_foreign_wait():
    """a caller that waits for the plan's stream itself (torch.cuda.synchronize here) asks pmt_plan_check for the device-side error state"""
    import gpu_util as g
    from parametron_jl_amd import _lib
    plan, (dA, db, xvar, res, rc), (r, n) = _large_run_plan(g)
    g.call("pmt_plan_update", plan); torch.cuda.synchronize()
    g.call("pmt_plan_check", plan)                                       # clean
    try:
        g.call("pmt_set_fault_injection", 2)
        g.call("pmt_plan_update", plan)
        torch.cuda.synchronize()
        with pytest.raises(_lib.HipError, match="grid barrier"):
            g.call("pmt_plan_check", plan)
    finally:
        g.call("pmt_set_fault_injection", 0)
    g.call("pmt_plan_check", plan)                                       # cleared by the report
    g.call("pmt_plan_update", plan); g.call("pmt_plan_synchronize", plan)
    g.call("pmt_plan_destroy", plan)


def test_toggling_fusion_does_not_grow_the_plan():
    """build_exec used to allocate a new node table per fused run at every rebuild and keep the old ones until pmt_plan_destroy"""
    import gpu_util as g
    plan, bufs, (r, n) = _large_run_plan(g)
    L = g.lib()
    L.pmt_plan_bytes_allocated.restype = C.c_size_t
    fused = int(L.pmt_plan_bytes_allocated(plan))
    g.call("pmt_plan_set_fusion", plan, 0)
    unfused = int(L.pmt_plan_bytes_allocated(plan))
    assert unfused < fused
    for _ in range(10):
        g.call("pmt_plan_set_fusion", plan, 1)
        assert int(L.pmt_plan_bytes_allocated(plan)) == fused
        g.call("pmt_plan_update", plan)
        g.call("pmt_plan_set_fusion", plan, 0)
        assert int(L.pmt_plan_bytes_allocated(plan)) == unfused
    g.call("pmt_plan_set_fusion", plan, 1)
    g.call("pmt_plan_update", plan); g.call("pmt_plan_synchronize", plan)
    g.call("pmt_plan_destroy", plan)


def test_model_update_entry_point_as_a_c_host_would_use_it():
    """pmt_model_update (csrc/modelrun.hip) driven through the C ABI alone, the way a Julia / C host does: README
    Example 1's shapes with HOST-updated A (Julia layout: column-major, stride (1, r)), a numpy C-order C block (stride (n, 1)), a
    device-regenerated d (seed word), the constraint's constants stored straight into a page-locked array.  Every solve is compared with
    the ORACLE's matvecmul! + vecsubtract! + update!(::MOI.VectorAffineFunction) byte for byte; the dirty mask skips unchanged mailboxes."""
    import gpu_util as g
    from oracle import oracle as O
    from parametron_jl_amd import _lib
    r, n = 6, 8
    L = g.lib()
    rng = np.random.default_rng(11)
    s = g.stream()
    plan = C.c_void_p()
    g.call("pmt_plan_create", 0, s, C.byref(plan))

    def pinned(count, dtype):
        p = C.c_void_p()
        g.call("pmt_host_alloc", count * np.dtype(dtype).itemsize, C.byref(p))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dtype).itemsize,)).view(dtype)
        a[...] = np.zeros(count, dtype=dtype)
        return a
    A_host = np.asfortranarray(rng.random((r, n)))                        # the host language's own array (Julia: Matrix{Float64})
    C_host = np.ascontiguousarray(rng.random((r, n)))                     # numpy C order
    b_host = rng.random(r)
    ldA = 8                                                               # device layout: padded leading dimension
    mbA, mbC, mbb = pinned(ldA * n, np.float64), pinned(ldA * n, np.float64), pinned(r, np.float64)
    dA, dC, db, dd = g.empty_f64(ldA * n), g.empty_f64(ldA * n), g.empty_f64(r), g.empty_f64(r)
    xvar = g.to_dev(np.arange(1, n + 1, dtype=np.int64))
    vmap_h = rng.permutation(n).astype(np.int64) + 5
    vmap = g.to_dev(vmap_h)
    t1, c1 = pinned(r * n, g.VAT), pinned(r, np.float64)                  # MOI.VectorAffineFunction of A*x - b: terms, constants (host memory)
    t2, c2 = pinned(r * n, g.VAT), pinned(r, np.float64)                  # ... of C*x - d
    seed = C.c_uint64(0)
    rec = C.c_void_p(L.pmt_plan_recording_stream(plan))
    devp = lambda a: C.c_void_p(a.ctypes.data)
    g.call("pmt_plan_begin_record", plan)
    g.call("pmt_copy_bytes", g.ptr(dA), devp(mbA), 8 * ldA * n, rec)      # mailbox -> Parameter buffer, inside the one launch
    g.call("pmt_copy_bytes", g.ptr(dC), devp(mbC), 8 * ldA * n, rec)
    g.call("pmt_copy_bytes", g.ptr(db), devp(mbb), 8 * r, rec)
    g.call("pmt_fill_uniform_dyn_f64", g.ptr(dd), r, 1, r, C.byref(seed), 2.0, rec)
    g.call("pmt_affine_pack_vector_f64", g.ptr(dA), ldA, r, n, g.ptr(xvar), g.ptr(db), -1, g.ptr(vmap), 0, devp(t1), devp(c1), rec)
    g.call("pmt_affine_pack_vector_f64", g.ptr(dC), ldA, r, n, g.ptr(xvar), g.ptr(dd), -1, g.ptr(vmap), 0, devp(t2), devp(c2), rec)
    g.call("pmt_plan_end_record", plan)
    groups, nodes, ln = C.c_int(), C.c_int(), C.c_int64()
    g.call("pmt_plan_fused", plan, C.byref(groups), C.byref(nodes), C.byref(ln))
    assert ln.value == 1                                                  # the whole update! is ONE launch
    model = C.c_void_p()
    g.call("pmt_model_create", plan, C.byref(model))
    sA, sC, sb, sd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    g.call("pmt_model_add_mailbox", model, devp(A_host), r, n, 1, r, devp(mbA), ldA, C.byref(sA))
    g.call("pmt_model_add_mailbox", model, devp(C_host), r, n, n, 1, devp(mbC), ldA, C.byref(sC))
    g.call("pmt_model_add_mailbox", model, devp(b_host), r, 0, 1, 0, devp(mbb), r, C.byref(sb))
    g.call("pmt_model_add_seed", model, C.byref(seed), C.c_uint64(4), C.c_uint64(1000), C.byref(sd))
    total = pinned(1, np.float64)
    g.call("pmt_model_add_constant", model, devp(c1), devp(total))        # (*dst = *src behind the wait: here c1[0] -> total[0])
    assert (sA.value, sC.value, sb.value, sd.value) == (0, 1, 2, 3) and L.pmt_model_num_slots(model) == 4
    xi = np.arange(1, n + 1, dtype=np.int64)

    def check(epoch_of_d, Av, Cv, bv):
        w1t, w1c = O.AffVec(r).vecsubtract(O.AffVec(r).matvecmul_vars(Av, xi), bv).moi(vmap_h)
        dv = O.fill_uniform(r, 4 + 1000 * epoch_of_d, 2.0)
        w2t, w2c = O.AffVec(r).vecsubtract(O.AffVec(r).matvecmul_vars(Cv, xi), dv).moi(vmap_h)
        assert np.array_equal(t1.view(np.int64), w1t.view(np.int64)) and np.array_equal(c1.view(np.int64), w1c.view(np.int64))
        assert np.array_equal(t2.view(np.int64), w2t.view(np.int64)) and np.array_equal(c2.view(np.int64), w2c.view(np.int64))
        assert total[0] == c1[0]
    g.call("pmt_model_update", model, None, 0, 1)                         # NULL mask: everything is dirty
    check(0, A_host, C_host, b_host)
    for it in range(1, 6):
        A_keep = A_host.copy()
        mask = (C.c_ubyte * 4)(0, 1, 1, 1)                                # A's byte is 0: the user did not touch it ...
        A_host[...] = rng.random((r, n))                                  # ... (this write is NOT announced: the mailbox keeps the old value)
        C_host[...] = rng.random((r, n)); b_host[...] = rng.random(r)
        g.call("pmt_model_update", model, mask, 4, 0)                     # asynchronous form
        g.call("pmt_model_wait", model)
        check(it, A_keep, C_host, b_host)
        A_host[...] = A_keep
    with pytest.raises(_lib.DimensionMismatch):
        g.call("pmt_model_update", model, (C.c_ubyte * 3)(1, 1, 1), 3, 1)
    g.call("pmt_model_destroy", model)
    g.call("pmt_plan_destroy", plan)
    for a in (mbA, mbC, mbb, t1, c1, t2, c2, total):
        g.call("pmt_host_free", C.c_void_p(a.ctypes.data))
