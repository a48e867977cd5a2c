"""The Julia glue (julia/QuantumPropagatorsHIPExt.jl) cannot be executed here (no Julia in the image), so
its binding to the C ABI is checked mechanically instead: every `ccall((:qp_..., LIB), Ret, (types...), ...)`
is parsed and compared -- name, return type, arity and every argument type -- with the prototype in
include/qprop.h; the Julia mirror of `qp_newton_stats` is compared field by field; every destructor named
in a `Handle(...)` exists; and the excerpt shown in INTEGRATION.md is regenerated from the file, so the
two cannot drift."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "QuantumPropagatorsHIPExt.jl")
HDR = os.path.join(ROOT, "include", "qprop.h")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")

OPAQUE = {"qp_ctx", "qp_matrix", "qp_operator", "qp_state", "qp_krylov", "qp_cheby", "qp_newton", "qp_split", "qp_comm",
          "qp_sharded_cheby"}
SCALARS = {"int": "int", "double": "double", "int64_t": "int64", "size_t": "size_t", "qp_c128": "c128", "unsigned": "uint",
           "int32_t": "int32", "uint64_t": "uint64", "char": "char", "qp_acc_defer": "struct:qp_acc_defer"}


def _strip_comments(txt):
    return re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)


def c_class(decl):
    """Canonical class of one C parameter / return declaration."""
    d = re.sub(r"\b(const|struct|restrict)\b", " ", decl)
    arr = "[" in d
    d = re.sub(r"\[[^\]]*\]", " ", d)
    d = d.strip()
    stars = d.count("*") + (1 if arr else 0)
    toks = re.sub(r"\*", " ", d).split()
    if not toks:
        raise ValueError(decl)
    base = toks[0]
    if base in ("qp_func_cb", "qp_exchange_cb"):
        return "ptr:void"                  # a C function pointer: @cfunction / C_NULL on the Julia side
    if stars == 0:
        return SCALARS[base]
    if base == "char":
        return "cstring" if stars == 1 else "ptr:ptr"
    if stars >= 2:
        return "ptr:ptr"
    if base == "void" or base in OPAQUE:
        return "ptr:void"
    if base in SCALARS:
        return "ptr:" + SCALARS[base]
    return "ptr:struct:" + base            # a by-reference struct of the header (qp_newton_stats, ...)


def parse_header():
    txt = _strip_comments(open(HDR).read())
    protos = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][A-Za-z_0-9]*\s*\**)\s*\b(qp_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if name.endswith("_cb"):
            continue
        params = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
        protos[name] = (c_class(ret), [c_class(a) for a in params], params)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(qp_[a-z_0-9]+)\s*;", txt, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            typ, names = stmt.rsplit(" ", 1)[0], stmt
            base = stmt.split()[0] if not stmt.startswith("const") else stmt.split()[1]
            for nm in stmt[stmt.index(base) + len(base):].split(","):
                try:
                    cls = c_class((("const " if stmt.startswith("const") else "") + base + " " + nm).strip())
                except KeyError:
                    cls = "?"
                fields.append((cls, nm.strip().lstrip("*")))
        structs[m.group(2)] = fields
    return protos, structs


JL_SCALARS = {"Cint": "int", "Cdouble": "double", "Int64": "int64", "Csize_t": "size_t", "C128": "c128", "ComplexF64": "c128",
              "Cstring": "cstring", "Cvoid": "void"}


JL_STRUCTS = {"NewtonStats": "qp_newton_stats"}      # Julia mirror -> C struct (fields compared below)


def jl_class(t):
    t = t.strip()
    if t in JL_SCALARS:
        return JL_SCALARS[t]
    m = re.fullmatch(r"Ptr\{(.*)\}", t)
    if not m:
        raise ValueError(f"unknown Julia ccall type {t!r}")
    inner = m.group(1).strip()
    if inner.startswith("Ptr{"):
        return "ptr:ptr"
    if inner == "Cvoid":
        return "ptr:void"
    if inner in JL_SCALARS:
        return "ptr:" + JL_SCALARS[inner]
    return "ptr:struct:" + JL_STRUCTS.get(inner, inner)


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def parse_ccalls():
    """[(name or None for a dynamic symbol, ret, [arg types], n actual args, line)] for every ccall in the glue."""
    txt = open(JL).read()
    txt_nc = "\n".join(re.sub(r"#.*$", "", ln) for ln in txt.splitlines())
    calls = []
    for m in re.finditer(r"ccall\(", txt_nc):
        i = m.end()
        depth, j = 1, i
        while depth:
            ch = txt_nc[j]
            depth += ch == "("
            depth -= ch == ")"
            j += 1
        parts = _split_top(txt_nc[i:j - 1])
        sym = re.match(r"\(\s*(:?[A-Za-z_.0-9]+)\s*,\s*LIB\s*\)", parts[0])
        assert sym, f"ccall target not of the form (:name, LIB): {parts[0]!r}"
        name = sym.group(1)[1:] if sym.group(1).startswith(":") else None
        ret = parts[1]
        atypes = [t for t in _split_top(parts[2].strip()[1:-1]) if t]
        calls.append((name, ret, atypes, len(parts) - 3, txt_nc[:m.start()].count("\n") + 1))
    return calls


def test_every_ccall_matches_the_header():
    protos, _ = parse_header()
    calls = parse_ccalls()
    named = [c for c in calls if c[0] is not None]
    assert len(named) >= 20, "the glue binds the whole step path"
    seen = set()
    for name, ret, atypes, nargs, line in named:
        assert name in protos, f"julia:{line}: {name} is not declared in include/qprop.h"
        cret, cargs, cdecl = protos[name]
        assert jl_class(ret) == cret, f"julia:{line}: {name} returns {ret}, header says {cret}"
        assert len(atypes) == len(cargs) == nargs, \
            f"julia:{line}: {name}: {len(atypes)} types / {nargs} values in the ccall, {len(cargs)} parameters in the header"
        for k, (jt, ct) in enumerate(zip(atypes, cargs)):
            assert jl_class(jt) == ct, f"julia:{line}: {name} argument {k + 1} ({cdecl[k]}): Julia type {jt} is {jl_class(jt)}, header wants {ct}"
        seen.add(name)
    # the entry points a propagator needs are all bound
    for must in ("qp_ctx_create", "qp_matrix_create", "qp_operator_create", "qp_operator_set_coeffs", "qp_state_create",
                 "qp_state_upload", "qp_state_download", "qp_cheby_coeffs", "qp_cheby_create", "qp_cheby_step", "qp_newton_create",
                 "qp_newton_step", "qp_specrange_arnoldi", "qp_copy", "qp_scal", "qp_axpy", "qp_dot", "qp_norm", "qp_fill", "qp_mul",
                 "qp_host_register", "qp_host_unregister", "qp_last_error"):
        assert must in seen, f"{must} is not bound by the glue"


def test_destructors_named_in_handles_exist():
    protos, _ = parse_header()
    txt = open(JL).read()
    names = set(re.findall(r"Handle\([^)]*?:(qp_[a-z_]+_destroy)", txt))
    assert {"qp_ctx_destroy", "qp_matrix_destroy", "qp_operator_destroy", "qp_state_destroy", "qp_cheby_destroy",
            "qp_newton_destroy"} <= names
    for n in names:
        assert n in protos and protos[n][0] == "int" and protos[n][1] == ["ptr:void"], n
    # the dynamic-symbol ccall of the finalizer has the destructor's shape
    dyn = [c for c in parse_ccalls() if c[0] is None]
    assert len(dyn) == 1 and jl_class(dyn[0][1]) == "int" and [jl_class(t) for t in dyn[0][2]] == ["ptr:void"]


def test_newton_stats_mirror_matches_field_by_field():
    _, structs = parse_header()
    cfields = structs["qp_newton_stats"]
    txt = open(JL).read()
    body = re.search(r"struct NewtonStats[^\n]*\n(.*?)\nend", txt, flags=re.S).group(1)
    jfields = []
    for stmt in re.split(r"[;\n]", body):
        stmt = stmt.split("#")[0].strip()
        if stmt:
            nm, ty = stmt.split("::")
            jfields.append((JL_SCALARS[ty.strip()], nm.strip()))
    assert jfields == cfields, (jfields, cfields)


def test_glue_has_a_device_resident_state_and_takes_every_generator_form():
    txt = open(JL).read()
    assert re.search(r"mutable struct HIPState <: AbstractVector\{ComplexF64\}", txt)
    for f in ("copyto!", "lmul!", "axpy!", "dot", "norm", "fill!"):
        assert re.search(rf"(Base|LinearAlgebra)\.{re.escape(f)}\([^)]*HIPState", txt), f
    # the in-place step path of a device-resident state has no transfer: _hand_back! returns first
    hb = re.search(r"function _hand_back!\(p\)\n(.*?)\nend", txt, flags=re.S).group(1)
    assert hb.strip().startswith("p.state isa HIPState && return p.state")
    step = re.search(r"function prop_step!\(p::ChebyHIPPropagator\)\n(.*?)\nend", txt, flags=re.S).group(1)
    assert "download" not in step and "Array(" not in step
    # Generator, tuple generator, static Operator and plain matrix (src/controls.jl:442-475) all dispatch
    for sig in ("lazy_sum(G::Generator)", "lazy_sum(O::Operator)", "lazy_sum(A::AbstractMatrix)", "lazy_sum(terms::Tuple)"):
        assert sig in txt, sig
    assert "init_prop(state, generator, tlist, ::Val{:ChebyHIP}" in txt and "init_prop(state, generator, tlist, ::Val{:NewtonHIP}" in txt


BEGIN, END = "<!-- BEGIN generated from julia/QuantumPropagatorsHIPExt.jl (tests/test_julia_glue_signatures.py) -->", \
    "<!-- END generated -->"


def excerpt():
    """The parts of the glue INTEGRATION.md shows: the handle type, the device state type and the two step functions."""
    txt = open(JL).read()

    def block(start_pat, end_pat="\nend\n"):
        i = re.search(start_pat, txt).start()
        j = txt.index(end_pat, i) + len(end_pat)
        return txt[i:j]
    parts = [block(r"mutable struct Handle\n"), block(r"mutable struct HIPState <: AbstractVector"),
             block(r"function LinearAlgebra\.axpy!\(a::Number, x::HIPState"),
             block(r"# What the reference accepts as a generator", "lazy_sum(G) = throw"),
             block(r"# prop_step!\(::ChebyPropagator\)"), block(r"function prop_step!\(p::NewtonHIPPropagator\)")]
    return "```julia\n" + "\n".join(p.rstrip("\n") + "\n" for p in parts).replace("lazy_sum(G) = throw", "") + "```\n"


def test_integration_md_excerpt_is_generated_from_the_file():
    md = open(INTEGRATION).read()
    assert BEGIN in md and END in md, "INTEGRATION.md lost its generated block markers"
    cur = md[md.index(BEGIN) + len(BEGIN):md.index(END)].strip("\n")
    want = excerpt().strip("\n")
    if cur != want and os.environ.get("QP_REGENERATE_DOCS") == "1":
        md = md[:md.index(BEGIN) + len(BEGIN)] + "\n" + want + "\n" + md[md.index(END):]
        with open(INTEGRATION, "w") as f:
            f.write(md)
        cur = want
    assert cur == want, "INTEGRATION.md's glue excerpt differs from julia/QuantumPropagatorsHIPExt.jl: run " \
                        "QP_REGENERATE_DOCS=1 python -m pytest tests/test_julia_glue_signatures.py"
