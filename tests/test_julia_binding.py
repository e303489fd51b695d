"""The Julia binding (julia/*.jl) cannot be executed here -- the image has no Julia -- so it is
checked statically: every `ccall((:name, lib), Ret, (types...), args...)` is parsed and compared
with the prototype of `name` in include/nsk.h (return type, number and C type of each argument,
number of values passed), the two structs are compared field by field with the header, and every
function the header declares must have a stub.  Types alone would let two same-typed scalars change
places (dx and dy, say), so the NAMES are compared too: a value whose name -- Greek letters spelled
out, field accessors dropped -- is the name of a parameter of the C function must be passed in that
parameter's place."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def header_prototypes():
    text = strip_comments(open(os.path.join(ROOT, "include", "nsk.h")).read())
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char \*|int )\s*(nsk_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        types = []
        for a in params:
            a = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a      # drop the parameter name
            types.append(" ".join(a.replace("*", " * ").split()))
        protos[name] = (ret, types)
    return protos


def header_parameter_names():
    text = strip_comments(open(os.path.join(ROOT, "include", "nsk.h")).read())
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    names = {}
    for m in re.finditer(r"(?:const char \*|int )\s*(nsk_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = " ".join(m.group(2).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        names[m.group(1)] = [re.search(r"(\w+)$", a).group(1) if re.search(r"\w$", a) else "" for a in params]
    return names


GREEK = {"τ": "tau", "μ": "mu", "ρ": "rho", "β": "b", "∇V": "divV", "∇": "div", "Δ": "d", "σ": "sigma", "ε": "eps", "α": "alpha"}


def spelled_out(value):
    """`τxx.p` -> `tauxx`, `dτ` -> `dtau`, `sinβ` -> `sinb`; anything that is not a plain name -> ''."""
    v = value.strip()
    v = re.sub(r"\.p$", "", v)
    for g, latin in GREEK.items():
        v = v.replace(g, latin)
    return v if re.fullmatch(r"[A-Za-z_]\w*", v) else ""


C_TO_JULIA = {
    "int": {"Cint"},
    "double": {"Cdouble"},
    "size_t": {"Csize_t"},
    "double *": {"Ptr{Cdouble}", "PD"},
    "const double *": {"Ptr{Cdouble}", "PD"},
    "double * *": {"Ptr{Ptr{Cdouble}}"},
    "int *": {"Ptr{Cint}"},
    "nsk_ctx *": {"Ptr{Cvoid}"},
    "nsk_ctx * *": {"Ptr{Ptr{Cvoid}}"},
    "nsk_slabs *": {"Ptr{Cvoid}"},
    "nsk_slabs * *": {"Ptr{Ptr{Cvoid}}"},
    "const int *": {"Ptr{Cint}"},
    "const void *": {"Ptr{Cvoid}"},
    "void *": {"Ptr{Cvoid}"},
    "const char *": {"Cstring"},
    "const nsk_ns_params *": {"Ref{Params}"},
    "const nsk_ns_state *": {"Ref{State}"},
    "const nsk_bub_params *": {"Ref{BubParams}"},
    "const nsk_bub_dns_state *": {"Ref{DnsState}"},
    "nsk_bub_code3_state *": {"Ref{Code3State}"},
}


def split_top(s):
    """Split at commas that are not inside (), [] or {}."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls():
    calls = []
    for f in sorted(os.listdir(os.path.join(ROOT, "julia"))):
        if not f.endswith(".jl"):
            continue
        text = open(os.path.join(ROOT, "julia", f)).read()
        text = re.sub(r"#[^\n]*", "", text)
        for m in re.finditer(r"ccall\(", text):
            i, depth = m.end(), 1
            while depth:                      # find the matching parenthesis
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            parts = split_top(text[m.end():i - 1])
            name = re.match(r"\(\s*:(\w+)\s*,\s*lib\s*\)", parts[0]).group(1)
            types = split_top(parts[2].strip()[1:-1])
            calls.append((f, name, parts[1], types, parts[3:]))
    return calls


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    assert len(protos) >= 55
    calls = julia_ccalls()
    assert len(calls) >= 55
    for f, name, ret, types, args in calls:
        assert name in protos, f"{f}: {name} is not declared in include/nsk.h"
        cret, ctypes_ = protos[name]
        assert ret == {"int": "Cint", "const char *": "Cstring"}[cret], (f, name, ret)
        assert len(types) == len(ctypes_), f"{f}: {name} takes {len(ctypes_)} arguments, the ccall lists {len(types)}"
        assert len(args) == len(types), f"{f}: {name}: {len(types)} types but {len(args)} values"
        for i, (jt, ct) in enumerate(zip(types, ctypes_)):
            assert jt in C_TO_JULIA[ct], f"{f}: {name} argument {i + 1} is `{ct}` in the header, `{jt}` in the ccall"


def test_named_values_are_passed_in_their_parameters_place():
    names = header_parameter_names()
    checked, functions = 0, set()
    for f, name, _, _, args in julia_ccalls():
        params = names[name]
        for i, a in enumerate(args):
            v = spelled_out(a)
            if v and params.count(v) == 1:
                assert params.index(v) == i, (f"{f}: {name}: `{a.strip()}` is passed as argument {i + 1} ({params[i]}), "
                                              f"but the header's `{v}` is argument {params.index(v) + 1}")
                checked += 1
                functions.add(name)
    # most scalar arguments carry the header's names (dx, dy, dt, rho, mu, ...): the check has teeth
    assert checked >= 250 and len(functions) >= 40, (checked, len(functions))


def test_the_name_check_sees_a_swap():
    names = {"nsk_x": ["ctx", "A", "dx", "dy", "nx"]}
    args = ["c.h", "A.p", "dy", "dx", "A.n1"]
    params = names["nsk_x"]
    wrong = [a for i, a in enumerate(args) if spelled_out(a) in params and params.index(spelled_out(a)) != i]
    assert wrong == ["dy", "dx"]
    assert spelled_out("τxx.p") == "tauxx" and spelled_out("dτ") == "dtau" and spelled_out("Pr.n1") == "" and spelled_out("∇V.p") == "divV"


def test_every_declared_function_has_a_stub():
    called = {name for _, name, _, _, _ in julia_ccalls()}
    missing = sorted(set(header_prototypes()) - called)
    assert not missing, missing


def test_structs_match_the_header():
    text = strip_comments(open(os.path.join(ROOT, "include", "nsk.h")).read())
    jl = re.sub(r"#[^\n]*", "", open(os.path.join(ROOT, "julia", "NSK.jl")).read())
    for cname, jname in (("nsk_ns_params", "Params"), ("nsk_ns_state", "State"), ("nsk_bub_params", "BubParams"),
                         ("nsk_bub_dns_state", "DnsState"), ("nsk_bub_code3_state", "Code3State")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, re.S).group(1)
        want = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                ty, rest = decl.split(None, 1)
                for n in rest.split(","):
                    n = n.strip()
                    want.append((n.lstrip("*"), {"int": "Cint", "double": "Ptr{Cdouble}" if n.startswith("*") else "Cdouble"}[ty]))
        jbody = re.search(r"struct %s\n(.*?)\nend" % jname, jl, re.S).group(1)
        got = [tuple(x.strip().split("::")) for x in re.split(r"[;\n]", jbody) if x.strip()]
        assert got == want, (jname, [(a, b) for a, b in zip(got, want) if a != b][:3])


def test_drivers_keep_the_reference_surface():
    """`runme(; n, do_vis, do_save)` (scripts/NavierStokes_Re6.jl:12), `main(; do_vis, do_save)`
    (scripts/DNS.jl:14), both grids of the projection solver, the .mat keys and file names."""
    run = open(os.path.join(ROOT, "julia", "runme_nsk.jl")).read()
    assert re.search(r"function runme\(; n=1, do_vis=true, do_save=false", run)
    assert "n*256-1" in run and "n*160-1" in run and "255" in run and ":NavierStokes" in run
    assert 'matwrite("out_vis/step_$it.mat"' in run
    for key in ('"Pr"', '"Vx"', '"Vy"', '"C"', '"dx"', '"dy"'):
        assert key in run
    dns = open(os.path.join(ROOT, "julia", "dns_nsk.jl")).read()
    assert re.search(r"function main\(; do_vis=true, do_save=false", dns)
    assert 'matwrite("out_dns/step_$k.mat"' in dns
    assert os.path.exists(os.path.join(ROOT, "julia", "code3_nsk.jl"))
