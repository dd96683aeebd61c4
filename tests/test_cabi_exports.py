"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/parametron_hip.h declares."""
import ctypes as C
import os
import re

import pytest

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    entry.build()
    from parametron_jl_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "parametron_hip.h")).read()
    declared = set(re.findall(r"\b(pmt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    raw = C.CDLL(lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "symbol %s missing from libparametron_hip.so" % name
        assert name in lib.SIGNATURES, "symbol %s has no ctypes signature" % name
    assert set(lib.SIGNATURES) <= declared


def test_term_layouts_match_julia_isbits(lib):
    assert lib.LT.itemsize == 16 and lib.QT.itemsize == 24 and lib.VAT.itemsize == 24
    assert lib.VAT.names == ("out", "coeff", "var")
    assert lib.load().pmt_version() >= 100


def test_argument_validation_needs_no_gpu(lib):
    # validation happens before any launch, so error mapping can be checked on a CPU-only box
    L = lib.load()
    with pytest.raises(lib.DimensionMismatch):
        lib.call("pmt_affine_assemble_f64", None, 1, 2, 2, None, None, 0, None, None, None)      # lda < rows
    with pytest.raises(lib.ArgumentError):
        lib.call("pmt_affine_assemble_f64", None, 2, 2, 2, None, None, 5, None, None, None)      # bad sign
    with pytest.raises(lib.DimensionMismatch):
        lib.call("pmt_quad_expand_f64", -1, None, 0, None, None, 0, None, 0, None, None, None, None, None)
    assert b"" != L.pmt_last_error()


def test_product_path_fails_loudly_without_gpu(lib):
    if lib.load().pmt_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(lib.ErrorException):
        lib.require_gpu()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "parametron.jl_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|libparametron_oracle|pmo_[a-z_]+\s*\(", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), "product source %s references the oracle" % os.path.join(dirpath, f)


def test_header_is_plain_c_and_layouts_hold_for_a_c_compiler(tmp_path):
    """include/parametron_hip.h must be consumable by a C compiler (the boundary a cgo / ccall / ctypes host binds): compile a
    translation unit that includes it as C11 and checks the struct layouts the Julia side relies on."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include <stddef.h>\n#include "parametron_hip.h"\n'
                   '_Static_assert(sizeof(pmt_linear_term) == 16 && offsetof(pmt_linear_term, var) == 8, "LinearTerm");\n'
                   '_Static_assert(sizeof(pmt_quadratic_term) == 24 && offsetof(pmt_quadratic_term, col) == 16, "QuadraticTerm");\n'
                   '_Static_assert(sizeof(pmt_vector_affine_term) == 24 && offsetof(pmt_vector_affine_term, coeff) == 8, "VectorAffineTerm");\n'
                   'int (*probe)(const double *, int64_t, int64_t, int64_t, const int64_t *, const double *, int, pmt_linear_term *, double *, void *)'
                   ' = pmt_affine_assemble_f64;\nint main(void) { return PMT_OK; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _c_param_class(decl):
    """class of a C parameter declaration: 'ptr' | 'i64' | 'int' | 'f64' | 'size' | 'u64' | 'cstr'"""
    d = re.sub(r"/\*.*?\*/", "", decl).strip()
    if "*" in d:
        return "cstr" if re.match(r"(const\s+)?char\s*\*", d) else "ptr"
    if re.search(r"\buint64_t\b", d):
        return "u64"
    if re.search(r"\bint64_t\b", d):
        return "i64"
    if re.search(r"\bsize_t\b", d):
        return "size"
    if re.search(r"\bdouble\b", d):
        return "f64"
    if re.search(r"\bint\b", d):
        return "int"
    raise AssertionError("unclassified C parameter %r" % decl)


def _julia_type_class(t):
    t = t.strip()
    if t in ("DevPtr", "Cstring") or t.startswith(("Ptr{", "Ref{")):
        return "cstr" if t == "Cstring" else "ptr"
    return {"Int64": "i64", "Cint": "int", "Cdouble": "f64", "Float64": "f64", "Csize_t": "size", "UInt64": "u64"}.get(t) or \
        (_ for _ in ()).throw(AssertionError("unclassified Julia ccall type %r" % t))


def test_julia_sources_match_the_header():
    """julia/*.jl cannot be executed here (no julia); keep them honest statically: every ccall names a declared symbol, passes exactly as
    many arguments as the C declaration has parameters, each of the matching TYPE CLASS (pointer / Int64 / Cint / Cdouble / Csize_t /
    UInt64 / Cstring), and declares the matching return type."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "parametron_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    decls = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(pmt_\w+)\s*\(([^;{]*?)\)\s*;", header):
        params = m.group(3).strip()
        plist = [] if params in ("", "void") else [x.strip() for x in params.split(",")]
        ret = m.group(1).strip()
        decls[m.group(2)] = ([_c_param_class(x) for x in plist], ret)
    ncalls = 0
    for fname in sorted(os.listdir(os.path.join(root, "julia"))):
        if not fname.endswith(".jl"):
            continue
        src = open(os.path.join(root, "julia", fname)).read()
        for m in re.finditer(r"ccall\(\(:(pmt_\w+), lib\),\s*([\w{}.]+),\s*\(", src):
            ncalls += 1
            name, jret = m.group(1), m.group(2)
            assert name in decls, "%s: %s is not declared in the header" % (fname, name)
            depth, i = 1, m.end()
            while depth:                                                       # the argument-type tuple, balanced parentheses
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            types = src[m.end():i - 1].strip().rstrip(",")
            tlist = [] if not types else [t for t in re.split(r",(?![^{]*})", types) if t.strip()]
            want, cret = decls[name]
            assert len(tlist) == len(want), "%s: %s has %d argument types in Julia, %d parameters in the header" % (fname, name, len(tlist), len(want))
            got = [_julia_type_class(t) for t in tlist]
            assert got == want, "%s: %s argument classes %r, header %r" % (fname, name, got, want)
            rclass = ("cstr" if "char" in cret else "ptr") if "*" in cret else _c_param_class(cret + " x")
            jclass = "ptr" if jret.startswith("Ptr{") else _julia_type_class(jret)
            assert jclass == rclass or (rclass == "cstr" and jclass in ("cstr", "ptr")), "%s: %s returns %s in Julia, %s in the header" % (fname, name, jret, cret)
    assert ncalls >= 30


def test_julia_backend_only_uses_bound_wrappers():
    """ParametronHIPBackend.jl calls the binding as `H.<name>(...)`: every such name must be defined in ParametronHIP.jl."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binding = open(os.path.join(root, "julia", "ParametronHIP.jl")).read()
    backend = open(os.path.join(root, "julia", "ParametronHIPBackend.jl")).read()
    defined = set(re.findall(r"^(?:function\s+)?([A-Za-z_!0-9]+)\(", binding, flags=re.M)) | set(re.findall(r"^(?:mutable\s+)?struct\s+(\w+)", binding, flags=re.M)) | {"DevPtr"}
    used = set(re.findall(r"\bH\.([A-Za-z_!0-9]+)", backend))
    assert used and used <= defined, "ParametronHIPBackend.jl uses unbound wrappers: %r" % sorted(used - defined)


def test_shipped_library_reads_no_environment_switches():
    """Tuning / ablation switches (PMT_GRAM_SK_ABLATE & co. could silently corrupt results in round 1) exist only in -DPMT_TUNING
    builds: the shipped library contains no PMT_* environment name and does not import getenv for them."""
    path = os.path.join(ROOT, "parametron.jl_amd", "lib", "libparametron_hip.so")
    if not os.path.exists(path):
        pytest.skip("library not built")
    blob = open(path, "rb").read()
    names = set(re.findall(rb"PMT_[A-Z_]{3,}", blob))
    assert not names, "environment switch names in the shipped library: %r" % sorted(names)
    for src in os.listdir(os.path.join(ROOT, "parametron.jl_amd", "csrc")):
        if src.endswith((".hip", ".h")):
            text = open(os.path.join(ROOT, "parametron.jl_amd", "csrc", src)).read()
            # every getenv must sit inside an #ifdef PMT_TUNING block
            depth_tuning, stack = 0, []
            for line in text.splitlines():
                t = line.strip()
                if t.startswith(("#ifdef", "#ifndef", "#if ")):
                    stack.append("PMT_TUNING" in t and t.startswith("#ifdef"))
                elif t.startswith("#else") and stack:
                    stack[-1] = False
                elif t.startswith("#endif") and stack:
                    stack.pop()
                if "getenv(" in line and not t.startswith("//"):
                    assert any(stack), "%s reads the environment outside #ifdef PMT_TUNING: %s" % (src, t)


def test_julia_update_path_has_no_allocating_constructors():
    """The reference promises `@allocated solve!(model) == 0` (README.md:8,138, test/model.jl:116-124).  The Julia backend cannot be run
    here, so its per-solve functions are scanned for constructs that allocate: array constructors, comprehensions, copies, conversions.
    (Setup functions — HIPModel, record_*, DeviceParameter — may allocate: they run once.)"""
    src = open(os.path.join(ROOT, "julia", "ParametronHIPBackend.jl")).read()
    per_solve = [r"function refresh!\(", r"function commit!\(", r"function fetch_A!\(", r"function copy_A!\(", r"function Parametron\.update!\(o::HIPObjective",
                 r"function Parametron\.update!\(c::HIPConstraint", r"function Parametron\.update!\(hm::HIPModel\)", r"function solve!\(hm::HIPModel\)",
                 r"function update_small!\(hm::HIPModel\)"]
    banned = [r"\bVector\{[^}]*\}\(", r"\bMatrix\{[^}]*\}\(", r"\bArray\{", r"\bzeros\(", r"\bones\(", r"\bcollect\(", r"\bcopy\(", r"\bsimilar\(",
              r"\b(?:Int64|Float64|Any)\[", r"\[[^\]\n]*\bfor\b[^\]\n]*\]", r"\bpush!\(", r"\bvcat\(", r"\bhcat\(", r"\bconvert\(", r"\bstring\("]
    checked = 0
    for head in per_solve:
        m = re.search(head, src)
        assert m, "per-solve function %s not found in the Julia backend" % head
        body, depth = [], 0
        for line in src[m.start():].splitlines():
            code = line.split("#")[0]
            depth += len(re.findall(r"\b(function|if|for|while|try|let|do|begin)\b", code)) - len(re.findall(r"\bend\b", code))
            body.append(code)
            if depth <= 0 and len(body) > 1:
                break
        text = "\n".join(body)
        for pat in banned:
            hit = re.search(pat, text)
            assert not hit, "%s allocates in the per-solve path: %r" % (head, hit.group(0))
        checked += 1
    assert checked == len(per_solve)


def test_julia_backend_is_strict_by_default():
    """A record whose shape the analysis does not know must not silently run the reference's CPU update! (a Julia user could not tell a GPU
    solve from a CPU solve): HIPModel(model) throws an ArgumentError by default, `strict = false` is the explicit opt-out, on_device(hm)
    reports which record runs where, and update!(hm) reaches the reference's update!(record, optimizer, varmap) only through cpu_update!,
    which it calls only when the model is not strict.  (Static: no julia in this image.)"""
    src = open(os.path.join(ROOT, "julia", "ParametronHIPBackend.jl")).read()
    assert re.search(r"function HIPModel\(model::Model;[^)]*strict::Bool = true", src)
    assert "export HIPModel, solve!, on_device" in src and re.search(r"function on_device\(hm::HIPModel\)", src)
    # both catch sites of an Unsupported record throw when strict, before the @info fallback
    sites = re.findall(r"err isa Unsupported \|\| rethrow\(\)(.*?)push!\(hm\.cpu_records", src, flags=re.S)
    assert len(sites) == 2
    for body in sites:
        assert body.index("strict && throw(unsupported_record(") < body.index("@info")
    assert "ArgumentError(" in src[src.index("unsupported_record(what"):src.index("unsupported_record(what") + 400]

    def body_of(head):
        m = re.search(head, src)
        assert m, head
        out, depth = [], 0
        for line in src[m.start():].splitlines():
            code = line.split("#")[0]
            depth += len(re.findall(r"\b(function|if|for|while|try|let|do|begin)\b", code)) - len(re.findall(r"\bend\b", code))
            out.append(code)
            if depth <= 0 and len(out) > 1:
                break
        return "\n".join(out)
    upd = body_of(r"function Parametron\.update!\(hm::HIPModel\)")
    assert "m.model_var_to_optimizer)" not in upd, "update!(hm) calls the reference's CPU update! directly"
    assert re.search(r"hm\.strict \|\| cpu_update!\(hm\)", upd)
    cpu = body_of(r"function cpu_update!\(hm::HIPModel\)")
    assert "Parametron.update!(r, m.optimizer, m.model_var_to_optimizer)" in cpu
    small = body_of(r"function update_small!\(hm::HIPModel\)")     # the small-model form of update!(hm): the same guard
    assert re.search(r"hm\.strict \|\| cpu_update!\(hm\)", small) and "m.model_var_to_optimizer)" not in small
    assert len(re.findall(r"cpu_update!\(hm\)", src)) == 2 == len(re.findall(r"hm\.strict \|\| cpu_update!\(hm\)", src))          # guarded calls only


def test_constant_order_is_host_arithmetic(lib):
    """pmt_quad_gram_constant_order: the fixed summation order of the node's constant follows from (rows, cols) alone — the fused tall forms' order up to 2048
    columns; beyond, sequential where the contraction hides the chain (config 2) or 2048 chains (long vectors, cost model); no GPU needed"""
    def order(r, n):
        o, g, s = C.c_int(), C.c_int(), C.c_int()
        lib.call("pmt_quad_gram_constant_order", r, n, C.byref(o), C.byref(g), C.byref(s))
        return o.value, g.value, s.value
    # config 2 and its neighbours (2049 .. 4096 columns, below 2^29 elements): the one-launch form on 64 x 64 tiles (round 6c: gram_mid_big), order 5
    assert order(4096, 4096) == (5, 1, 512) and order(1000, 2049)[0] == 5 and order(8192, 4096)[0] == 5 and order(4096, 2304)[0] == 5
    assert order(8193, 4096)[0] == 5 and order(65536, 4096)[0] == 5 and order(131072, 2560)[0] == 5
    assert order(8, 8) == (0, 1, 0) and order(64, 1)[0] == 0 and order(256, 1)[0] == 3   # tiny shapes (the small-plan node, at most 64 rows): sequential
    # beyond (more rows, more columns): the stream-K node, the constant by the cost model (sequential where the contraction hides its chain)
    assert order(131072, 4096) == (1, 2048, 0) and order(4096, 4097)[0] == 0 and order(1000, 5000)[0] == 0 and order(16384, 4100)[0] == 1
    # up to 2048 columns the fused tall forms, whatever the row count: one tile (order 2; 3 = sixteen row-pair lanes, <= 16 columns) ..
    assert order(80, 96) == (2, 3, 32) and order(1000, 128) == (2, 32, 32) and order(4096, 128) == (2, 64, 32)
    o, g, s = order(1 << 20, 128)
    assert (o, s) == (2, 32) and g == 512
    o, g, s = order(8192, 128)
    assert o == 2 and g == 8192 // 64                           # at least 64 rows per workgroup
    # narrow panels (<= 64 columns): below 32768 rows the panel kernel (32 columns x 128-row stages: order 2; 16 x 256: order 3), from there the
    # stream form (order 4) — iterations of 32 (64 columns: 16) rows dealt out to the waves of <= 256 (64 columns: 512) workgroups
    assert order(5000, 17) == (2, 40, 128) and order(1024, 1) == (3, 4, 256) and order(100, 40)[0] == 2
    assert order(16384, 17)[0] == 2 and order(32768, 17) == (4, 256, 32) and order(77777, 50) == (4, 512, 16) and order(33001, 9)[0] == 4
    assert order(1 << 20, 64) == (4, 512, 16) and order(1 << 20, 16) == (4, 256, 32) and order(1 << 20, 32) == (4, 256, 32)
    assert order(1 << 20, 48) == (4, 512, 32) and order(40000, 40) == (4, 313, 32) and order(20000, 40)[0] == 2      # 33 .. 48 columns: three column groups (stream form only)
    # .. or several.  Most of them take the one-launch form on 64 x 64 tiles (gram_mid.hip; the constant by one more workgroup of that launch, order 5:
    # 512 strided chains) — gram.hip: gram_mid_applies, by measured times: up to 192 columns 2^25 elements, below 320 columns 2^26, below 384
    # columns 2^27, from 384 columns everything the fast load path reaches (below 2^29 elements)
    assert order(300, 300) == (5, 1, 512) and order(4096, 512) == (5, 1, 512) and order(100, 1000)[0] == 5 and order(8192, 256)[0] == 5
    assert order(4096, 256)[0] == 5 and order(70, 130)[0] == 5 and order(4097, 512)[0] == 5 and order(17, 130)[0] == 5 and order(31, 300)[0] == 5
    assert order(16384, 1024)[0] == 5 and order(131072, 256)[0] == 5 and order(65536, 1024)[0] == 5 and order(1024, 2048)[0] == 5 and order(2048, 1537)[0] == 5
    assert order(262144, 512)[0] == 5 and order(524288, 512)[0] == 5 and order(65536, 2048)[0] == 5 and order(8192, 2048)[0] == 5 and order(2049, 1537)[0] == 5
    assert order(230000, 256)[0] == 5 and order(160000, 320)[0] == 5 and order(1 << 20, 384)[0] == 5 and order(100000, 129)[0] == 5
    # .. the rest the diagonal tiles' kernel + the strict stream-K launch (tile 0's workgroups the constant, order 2): few, narrow panels out of HBM
    assert order(262144, 256)[0] == 2 and order(524288, 129)[0] == 2 and order(1 << 20, 192)[0] == 2 and order(786432, 320)[0] == 2
    assert order(1 << 20, 512)[0] == 2 and order(380000, 129)[0] == 2 and order(300000, 160)[0] == 2
    with pytest.raises(lib.ArgumentError):
        lib.call("pmt_quad_gram_constant_order", -1, 4, None, None, None)


def test_batch_shard_is_host_arithmetic(lib):
    per, first = C.c_int64(), C.c_int64()
    lib.call("pmt_batch_shard", 8192, 8, 3, C.byref(per), C.byref(first))
    assert (per.value, first.value) == (1024, 3072)
    with pytest.raises(lib.DimensionMismatch):
        lib.call("pmt_batch_shard", 8191, 8, 0, C.byref(per), C.byref(first))


def test_tall_kernel_dealings_in_the_source_are_the_generator_s():
    """csrc/gram_tall.hip's `tw_*` tables and TALL_BLOCKS are pasted from tools/gen_tall_deal.py: the source must hold exactly what the generator
    emits, every dealing must cover the upper triangle of its NBC x NBC block grid once, and the waves' MFMA counts must be level"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_tall_deal as G
    src = open(os.path.join(ROOT, "parametron.jl_amd", "csrc", "gram_tall.hip")).read()
    T = G.tables()
    order = [5, 6, 7, 8]
    for name in ("nblk", "nr", "nc", "row", "col", "blk", "blocks"):
        text = G.braces([T[n][name] for n in order])
        assert text in src, "gram_tall.hip does not hold the generator's `%s` table" % name
    for nbc, waves in G.DEALS.items():
        blocks = sorted(b for w in waves for b in w)
        assert blocks == sorted((a, b) for b in range(nbc) for a in range(b + 1))
        mf = [sum(3 if a == b else 4 for a, b in w) for w in waves]
        assert max(mf) - min(mf) <= 4 and max(len(w) for w in waves) <= {5: 4, 6: 6, 7: 7, 8: 9}[nbc]
