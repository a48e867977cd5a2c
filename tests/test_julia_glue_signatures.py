"""The Julia glue (julia/QuantumPropagatorsHIPExt.jl) cannot be executed here (no Julia in the image), so
its binding to the C ABI is checked mechanically instead: every `ccall((:qp_..., LIB), Ret, (types...), ...)`
is parsed and compared -- name, return type, arity and every argument type -- with the prototype in
include/qprop.h (a `ccall` whose target is not a LITERAL `(:qp_name, LIB)` pair fails the test: Julia rejects a symbol
resolved at run time); the Julia mirror of `qp_newton_stats` is compared field by field; every destructor handed to a
`Handle(...)` is a one-line `ccall` of an existing destructor; every helper the glue imports from QuantumPropagators
exists in the reference with the arity and keywords used (tests/golden/reference_api.json, extracted from the reference's
sources by tests/golden/make_reference_api.py); the propagator structs carry the fields the reference's helpers and
`check_propagator` touch; and the excerpt shown in INTEGRATION.md is regenerated from the file, so the two cannot drift."""
import json
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
              "Cstring": "cstring", "Cvoid": "void", "UInt64": "uint64"}


JL_STRUCTS = {"NewtonStats": "qp_newton_stats", "PauliString": "qp_pauli_string"}      # Julia mirror -> C struct (fields compared below)


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
    if inner == "UInt8":          # a caller-owned character buffer the library writes into (char* text, size_t len)
        return "cstring"
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
    """[(name, ret, [arg types], n actual args, line)] for every ccall in the glue."""
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
        line = txt_nc[:m.start()].count("\n") + 1
        # `ccall` wants its (name, library) pair as a constant expression: a literal Symbol and the constant LIB.  A field,
        # a variable or a call in the first position compiles to "first argument not a pointer or valid constant expression"
        sym = re.fullmatch(r"\(\s*:(qp_[a-z_0-9]+)\s*,\s*LIB\s*\)", parts[0])
        assert sym, f"julia:{line}: ccall target is not a literal (:qp_name, LIB) pair: {parts[0]!r}"
        name = sym.group(1)
        ret = parts[1]
        atypes = [t for t in _split_top(parts[2].strip()[1:-1]) if t]
        calls.append((name, ret, atypes, len(parts) - 3, line))
    return calls


def test_every_ccall_matches_the_header():
    protos, _ = parse_header()
    calls = parse_ccalls()
    named = calls
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
    # the handle stores a FUNCTION (one literal ccall per destructor), never a Symbol to be resolved at run time
    assert re.search(r"mutable struct Handle\n\s*ptr::Ptr\{Cvoid\}\n\s*destroy::Function\n", txt)
    assert not re.search(r"destroy::Symbol", txt)
    defs = dict(re.findall(r"^(_[a-z]+_destroy)\(p::Ptr\{Cvoid\}\) = ccall\(\(:(qp_[a-z_]+_destroy), LIB\), Cint, \(Ptr\{Cvoid\},\), p\)$",
                           txt, flags=re.M))
    used = set(re.findall(r"Handle\([^()]*?\b(_[a-z]+_destroy)\b", txt))
    assert used == set(defs), (used, set(defs))
    assert set(defs.values()) == {"qp_ctx_destroy", "qp_matrix_destroy", "qp_operator_destroy", "qp_state_destroy", "qp_cheby_destroy",
                                  "qp_newton_destroy"}
    for n in defs.values():
        assert n in protos and protos[n][0] == "int" and protos[n][1] == ["ptr:void"], n
    # every Handle(...) construction passes one of them (no Symbol literal left over)
    for m in re.finditer(r"\bHandle\(([^()]*(?:\([^()]*\))?[^()]*)\)", txt):
        args = _split_top(m.group(1))
        if len(args) >= 2 and not args[0].startswith("ptr::"):
            assert args[1] in defs, f"Handle(...) with destructor {args[1]!r}"


# ---- the reference side of the glue: imported helpers, struct fields, the propagator contract ---------------------
API = os.path.join(ROOT, "tests", "golden", "reference_api.json")
REFERENCE = "/root/reference"


def _glue_code():
    return "\n".join(re.sub(r"#.*$", "", ln) for ln in open(JL).read().splitlines())


def _calls_of(name, txt):
    """[(n positional, [keyword names], line)] for every CALL of `name` in the glue (definitions excluded)."""
    out = []
    for m in re.finditer(r"(?<![A-Za-z_0-9!.])" + re.escape(name) + r"\(", txt):
        line_start = txt.rfind("\n", 0, m.start()) + 1
        head = txt[line_start:m.start()]
        i = m.end() - 1
        depth, j = 0, i
        while True:
            depth += txt[j] in "([{"
            depth -= txt[j] in ")]}"
            j += 1
            if depth == 0:
                break
        after = txt[j:j + 40]
        if head.strip() in ("function", "") and (head.strip() == "function" or re.match(r"\s*=(?!=)", after)):
            continue                                   # a method definition of the glue itself
        inner = txt[i + 1:j - 1]
        semi = [k for k, ch in enumerate(inner) if ch == ";" and inner[:k].count("(") == inner[:k].count(")")]
        pos_txt, kw_txt = (inner[:semi[0]], inner[semi[0] + 1:]) if semi else (inner, "")
        pos = [a for a in _split_top(pos_txt) if a]
        kws = [re.split(r"[=\s]", k, maxsplit=1)[0] for k in _split_top(kw_txt) if k and not k.endswith("...")]
        # `f(a; kw = v)` may also be written `f(a, kw = v)`
        kws += [re.split(r"[=\s]", a, maxsplit=1)[0] for a in pos if re.match(r"[A-Za-z_]\w*\s*=(?!=)", a)]
        pos = [a for a in pos if not re.match(r"[A-Za-z_]\w*\s*=(?!=)", a)]
        splat = any(a.endswith("...") for a in pos)
        out.append((None if splat else len(pos), kws, txt[:m.start()].count("\n") + 1))
    return out


def test_reference_api_fixture_is_current():
    """tests/golden/reference_api.json is what make_reference_api.py extracts from the reference (checked where the
    reference is present: this container; the GPU box only has the committed file)."""
    import pytest
    if not os.path.isdir(os.path.join(REFERENCE, "src")):
        pytest.skip("the reference sources are not on this machine")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_api", os.path.join(ROOT, "tests", "golden", "make_reference_api.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert json.loads(json.dumps(mod.scan(REFERENCE), sort_keys=True)) == json.load(open(API))


def test_helpers_imported_from_the_reference_exist_with_the_arity_used():
    api = json.load(open(API))["functions"]
    txt = _glue_code()
    imported = set()
    for m in re.finditer(r"^(?:using|import) QuantumPropagators(?:\.[A-Za-z]+)?:\s*((?:[^\n]|\n[ \t]+\S)*)", txt, flags=re.M):
        imported |= {t.strip() for t in m.group(1).replace("\n", " ").split(",") if t.strip()}
    helpers = {n for n in imported if n[0] == "_" or n[0].islower()}
    assert {"_pwc_process_parameters", "_pwc_advance_time!", "_pwc_set_t!", "_get_uniform_dt", "get_controls", "discretize",
            "evaluate", "hamiltonian", "init_prop", "prop_step!", "reinit_prop!", "set_state!", "set_t!"} <= helpers
    for name in sorted(helpers):
        assert api.get(name), f"{name} is imported by the glue but the reference defines no such function"
        for npos, kws, line in _calls_of(name, txt):
            ok = False
            for sig in api[name]:
                arity = npos is None or (sig["min_positional"] <= npos and (sig["max_positional"] is None or npos <= sig["max_positional"]))
                kw = all(k in sig["keywords"] or sig["keyword_rest"] for k in kws)
                ok = ok or (arity and kw)
            assert ok, f"julia:{line}: {name} called with {npos} positional arguments and keywords {kws}: no such method in the reference " \
                       f"({[(s['at'], s['min_positional'], s['max_positional'], s['keywords']) for s in api[name]][:4]})"
    # the fully qualified extension of Interfaces.supports_inplace
    assert re.search(r"QuantumPropagators\.Interfaces\.supports_inplace\(::Type\{HIPState\}\) = true", txt)
    assert any(s["min_positional"] == 1 for s in api["supports_inplace"])


def test_method_extensions_have_the_reference_generic_shape():
    api = json.load(open(API))["functions"]
    txt = _glue_code()
    # init_prop(state, generator, tlist, ::Val{:X}; kwargs..., _...) -- four positional arguments, trailing keyword rest so that
    # propagate() may pass keywords of other layers through (src/cheby_propagator.jl:87-104)
    inits = re.findall(r"^function init_prop\(([^;]*);((?:.|\n)*?)\)\n", txt, flags=re.M)
    assert len(inits) == 2
    ref = [s for s in api["init_prop"] if s["at"].startswith("src/cheby_propagator.jl")][0]
    for pos, kw in inits:
        assert len(_split_top(pos)) == 4 == ref["min_positional"]
        assert _split_top(kw)[-1].endswith("..."), "init_prop must swallow unknown keywords"
    cheby_kw = {re.split(r"[=\s]", k, maxsplit=1)[0] for k in _split_top(inits[0][1]) if not k.endswith("...")}
    assert set(ref["keywords"]) <= cheby_kw, set(ref["keywords"]) - cheby_kw
    newton_ref = [s for s in api["init_prop"] if s["at"].startswith("src/newton_propagator.jl")][0]
    newton_kw = {re.split(r"[=\s]", k, maxsplit=1)[0] for k in _split_top(inits[1][1]) if not k.endswith("...")}
    assert set(newton_ref["keywords"]) <= newton_kw, set(newton_ref["keywords"]) - newton_kw
    for name, npos in (("prop_step!", 1), ("set_state!", 2), ("set_t!", 2), ("reinit_prop!", 2)):
        defs = re.findall(r"^(?:function )?" + re.escape(name) + r"\(([^;)]*)", txt, flags=re.M)
        assert defs, name
        for d in defs:
            assert len(_split_top(d)) == npos, (name, d)
            assert any(s["min_positional"] == npos for s in api[name])


def test_propagator_structs_carry_the_fields_the_reference_touches():
    api = json.load(open(API))
    txt = _glue_code()
    need = set(api["public_properties"]["names"])                               # src/propagator.jl:119-126 (check_propagator reads them)
    for fn in ("_pwc_set_t!", "_pwc_advance_time!"):                            # the helpers the glue calls
        need |= set(api["pwc_helper_fields"]["by_function"][fn])
    need |= {"generator", "controls", "n"}                                      # PWC conventions, src/pwc_utils.jl:5-24
    for struct in ("ChebyHIPPropagator", "NewtonHIPPropagator"):
        m = re.search(r"mutable struct " + struct + r"\{GT\} <: PWCPropagator\n((?:.|\n)*?)\nend", txt)
        assert m, struct
        fields = {re.split(r"::|\s", ln.strip().removeprefix("const ").strip(), maxsplit=1)[0] for ln in m.group(1).split("\n") if ln.strip()}
        assert need <= fields, (struct, need - fields)
    assert api["abstract_types"]["PWCPropagator"]["decl"] == "abstract type PWCPropagator <: PiecewisePropagator end"
    # Generator / Operator fields the glue reads
    for s, fields in (("Generator", ("ops", "amplitudes")), ("Operator", ("ops", "coeffs"))):
        assert set(fields) <= set(api["structs"][s]["fields"])
    for f in re.findall(r"\bG\.(\w+)\.\.\.", txt):
        assert f in api["structs"]["Generator"]["fields"]
    for f in re.findall(r"\bO\.(\w+)\.\.\.", txt):
        assert f in api["structs"]["Operator"]["fields"]


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


def test_pauli_string_mirror_matches_field_by_field():
    """julia/QuantumPropagatorsHIPExt.jl: struct PauliString is passed to qp_pauli_operator_create as an array of qp_pauli_string."""
    _, structs = parse_header()
    cfields = structs["qp_pauli_string"]
    txt = open(JL).read()
    body = re.search(r"struct PauliString[^\n]*\n(.*?)\nend", txt, flags=re.S).group(1)
    jfields = []
    for stmt in re.split(r"[;\n]", body):
        stmt = stmt.split("#")[0].strip()
        if stmt:
            nm, ty = stmt.split("::")
            jfields.append((JL_SCALARS[ty.strip()], nm.strip()))
    assert [n for _, n in jfields] == [n for _, n in cfields], (jfields, cfields)
    assert [t for t, _ in jfields] == [t for t, _ in cfields], (jfields, cfields)


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
    parts = [block(r"# One destructor per kind of handle", "        return h\n    end\nend\n"), block(r"mutable struct HIPState <: AbstractVector"),
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
