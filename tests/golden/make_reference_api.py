#!/usr/bin/env python3
"""Extract, from the reference's Julia sources, the signatures the Julia glue (julia/QuantumPropagatorsHIPExt.jl) relies
on: every method of the helper functions it imports (positional arity range, keyword names, varargs), the fields of the
structs it reads, the public property names of a propagator, and where each lives (file:line).  Written as DATA
(tests/golden/reference_api.json); tests/test_julia_glue_signatures.py checks the glue against it, and -- where
/root/reference is present -- that the file is what this script produces from the sources there.

    python tests/golden/make_reference_api.py [/root/reference]
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_api.json")

FUNCTIONS = ["_pwc_process_parameters", "_pwc_advance_time!", "_pwc_set_t!", "_get_uniform_dt", "get_controls", "discretize",
             "evaluate", "hamiltonian", "supports_inplace", "init_prop", "prop_step!", "reinit_prop!", "set_state!", "set_t!"]
STRUCTS = ["Generator", "Operator"]
ABSTRACT = ["PWCPropagator", "PiecewisePropagator", "AbstractPropagator"]


def balanced(txt, i):
    """txt[i] == '(' -> index just past the matching ')'."""
    depth, j = 0, i
    while True:
        ch = txt[j]
        depth += ch in "([{"
        depth -= ch in ")]}"
        j += 1
        if depth == 0:
            return j


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "([{"
        depth -= ch in ")]}"
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def signature(argtxt):
    """'(a, b::T = 1, c...; k = 2, _...)' without the outer parens -> dict."""
    parts = split_top(argtxt, ";")
    pos = split_top(parts[0]) if parts and not argtxt.lstrip().startswith(";") else []
    kws = split_top(parts[1]) if len(parts) > 1 else (split_top(parts[0]) if argtxt.lstrip().startswith(";") else [])
    nmin = nmax = 0
    varargs = False
    for p in pos:
        if re.search(r"\.\.\.\s*$", p):
            varargs = True
            continue
        nmax += 1
        if "=" not in re.sub(r"\{[^}]*\}", "", p):
            nmin += 1
    kwnames, kwrest = [], False
    for k in kws:
        if k.endswith("..."):
            kwrest = True
        else:
            kwnames.append(re.split(r"[:=\s]", k, maxsplit=1)[0])
    return {"min_positional": nmin, "max_positional": None if varargs else nmax, "keywords": sorted(kwnames), "keyword_rest": kwrest}


def strip_docstrings_and_comments(txt):
    txt = re.sub(r'"""(?:.|\n)*?"""', lambda m: "\n" * m.group(0).count("\n"), txt)       # keep the line numbers
    return "\n".join(re.sub(r"#.*$", "", ln) for ln in txt.split("\n"))


def scan(ref):
    src = os.path.join(ref, "src")
    api = {"functions": {f: [] for f in FUNCTIONS}, "structs": {}, "abstract_types": {}, "public_properties": None}
    for dirpath, _, files in sorted(os.walk(src)):
        for fn in sorted(files):
            if not fn.endswith(".jl"):
                continue
            path = os.path.join(dirpath, fn)
            rel = os.path.relpath(path, ref)
            txt = strip_docstrings_and_comments(open(path).read())
            for name in FUNCTIONS:
                for m in re.finditer(r"^(?:function\s+)?(?:[A-Za-z_.]+\.)?" + re.escape(name) + r"\(", txt, flags=re.M):
                    i = m.end() - 1
                    j = balanced(txt, i)
                    rest = txt[j:j + 200]
                    is_def = m.group(0).lstrip().startswith("function") or re.match(r"\s*(where\s*\{[^}]*\}\s*)?=(?!=)", rest)
                    if not is_def:
                        continue
                    sig = signature(txt[i + 1:j - 1])
                    sig["at"] = f"{rel}:{txt[:m.start()].count(chr(10)) + 1}"
                    api["functions"][name].append(sig)
            for s in STRUCTS:
                m = re.search(r"^(?:mutable\s+)?struct\s+" + s + r"\b[^\n]*\n((?:.|\n)*?)\n\s*(?:function|end)\b", txt, flags=re.M)
                if m:
                    fields = [re.split(r"::|\s", ln.strip(), maxsplit=1)[0] for ln in m.group(1).split("\n") if ln.strip()]
                    api["structs"][s] = {"fields": [f for f in fields if re.fullmatch(r"[A-Za-z_]\w*", f)],
                                         "at": f"{rel}:{txt[:m.start()].count(chr(10)) + 1}"}
            for a in ABSTRACT:
                m = re.search(r"^abstract type\s+" + a + r"\b[^\n]*", txt, flags=re.M)
                if m:
                    api["abstract_types"][a] = {"decl": " ".join(m.group(0).split()), "at": f"{rel}:{txt[:m.start()].count(chr(10)) + 1}"}
            if rel.endswith("pwc_utils.jl"):     # fields each _pwc_* helper reads / writes on the propagator it is given
                api["pwc_helper_fields"] = {"at": rel, "by_function": {}}
                for fm in re.finditer(r"^function (_pwc_[a-z_!]+)\((?:.|\n)*?\nend\b", txt, flags=re.M):
                    body = fm.group(0)
                    names = set(re.findall(r"(?:get|set)field!?\(propagator,\s*:(\w+)", body)) | set(re.findall(r"\bpropagator\.(\w+)", body))
                    api["pwc_helper_fields"]["by_function"][fm.group(1)] = sorted(names)
            m = re.search(r"public_properties\s*=\s*\(([^)]*)\)", txt)
            if m and rel.endswith("propagator.jl") and "interfaces" not in rel:
                api["public_properties"] = {"names": [t.strip().lstrip(":") for t in m.group(1).split(",") if t.strip()],
                                            "at": f"{rel}:{txt[:m.start()].count(chr(10)) + 1}"}
    return api


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    api = scan(ref)
    with open(OUT, "w") as f:
        json.dump(api, f, indent=1, sort_keys=True)
        f.write("\n")
    print(OUT, {k: len(v) for k, v in api["functions"].items()})


if __name__ == "__main__":
    main()
