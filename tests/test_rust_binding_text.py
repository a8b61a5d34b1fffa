"""The Rust side of the boundary exists only as text (there is no Rust toolchain or bindgen where
this is built), so that text IS the binding a maintainer would paste.  This test parses its `extern "C"` blocks and
`#[repr(C)]` structs and the prototypes and structs of include/rt_abi.h, and compares them mechanically: function
names, return types, parameter names, order and types; struct field names, order and types; and the struct sizes the
C layout rules give against the ctypes structs the Python binding uses.  Renaming, re-ordering or re-typing a
parameter or field on one side only fails here."""
import ctypes
import os
import re

from raytracer_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "float": "f32",
             "uint8_t": "u8", "void": "c_void", "char": "c_char"}
SIZES = {"i32": 4, "u32": 4, "i64": 8, "u64": 8, "f64": 8, "f32": 4, "u8": 1}


def c_type(text):
    """'const rt2d_scene *' -> '*const rt2d_scene'; 'rt_ctx **' -> '*mut *mut rt_ctx'; 'int32_t' -> 'i32'."""
    text = text.strip()
    stars = text.count("*")
    base = text.replace("*", " ").split()
    const = "const" in base
    base = [b for b in base if b not in ("const", "struct")]
    assert len(base) == 1, text
    t = C_SCALARS.get(base[0], base[0])
    for k in range(stars):
        # `const T *`: the const binds to the pointee of the innermost pointer
        t = ("*const " if (const and k == 0) else "*mut ") + t
    return t


def split_decl(decl):
    """'const double *ox' -> ('*const f64', 'ox'); 'double cam_pos[3]' -> ('[f64; 3]', 'cam_pos')."""
    decl = decl.strip()
    m = re.match(r"^(.*?)(\w+)\s*(\[(\d+)\])?$", decl, re.S)
    ty, name, arr = m.group(1), m.group(2), m.group(4)
    t = c_type(ty)
    return (f"[{t}; {arr}]" if arr else t), name


def parse_header():
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            # 'const double *a, *b' and 'int32_t a, b' declare several fields of one base type
            first, *rest = [s.strip() for s in stmt.split(",")]
            t0, n0 = split_decl(first)
            fields.append((n0, t0))
            base = re.match(r"^(.*?)\**\s*\w+\s*(\[\d+\])?$", first, re.S).group(1)
            for r in rest:
                t, n = split_decl(base + " " + r)
                fields.append((n, t))
        structs[m.group(3)] = fields
    funcs = {}
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    for m in re.finditer(r"([\w\s]+?\**)\s*\b(rt[23]?d?_\w+)\s*\(([^)]*)\)\s*;", body):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [split_decl(a)[::-1] for a in args.split(",")]
        funcs[name] = (c_type(ret), params)
    return structs, funcs


def rust_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return "\n".join(re.findall(r"```rust\n(.*?)```", text, re.S))


def norm_rust_type(t):
    return re.sub(r"\s+", " ", t.strip())


def split_top(s):
    """Split on commas that are not inside brackets."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "[(<":
            depth += 1
        if ch in "])>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def parse_rust():
    text = re.sub(r"//[^\n]*", " ", rust_blocks())
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\][^\n]*?\s*pub\s+struct\s+(\w+)\s*\{(.*?)\}", text, re.S):
        fields = []
        for f in split_top(m.group(2)):
            f = f.strip()
            if not f:
                continue
            name, ty = f.split(":", 1)
            fields.append((name.replace("pub", "").strip(), norm_rust_type(ty)))
        structs[m.group(1)] = fields
    funcs = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, re.S):
        for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", blk.group(1), re.S):
            params = []
            for a in split_top(m.group(2)):
                if a.strip():
                    n, t = a.split(":", 1)
                    params.append((n.strip(), norm_rust_type(t)))
            funcs[m.group(1)] = (norm_rust_type(m.group(3) or "()"), params)
    return structs, funcs


def test_every_function_is_bound_with_the_same_signature():
    _, cf = parse_header()
    _, rf = parse_rust()
    assert len(cf) >= 37, "the header parser lost most of the prototypes"
    assert sorted(cf) == sorted(rf), f"only in the header: {sorted(set(cf) - set(rf))}; only in the Rust text: {sorted(set(rf) - set(cf))}"
    for name, (ret, params) in cf.items():
        rret, rparams = rf[name]
        assert ret == rret, f"{name}: returns {ret} in the header, {rret} in the Rust text"
        assert [p[0] for p in params] == [p[0] for p in rparams], f"{name}: parameter names / order differ: {params} vs {rparams}"
        for (pn, pt), (_, rt) in zip(params, rparams):
            assert pt == rt, f"{name}({pn}): {pt} in the header, {rt} in the Rust text"


def c_layout_size(fields, structs):
    """Size of a C struct under the natural-alignment rules both ABIs share."""
    off, align = 0, 1

    def size_align(t):
        m = re.match(r"^\[(.+); (\d+)\]$", t)
        if m:
            s, a = size_align(m.group(1))
            return s * int(m.group(2)), a
        if t.startswith("*"):
            return 8, 8
        if t in SIZES:
            return SIZES[t], SIZES[t]
        return c_layout_size(structs[t], structs), max(size_align(ft)[1] for _, ft in structs[t])

    for _, t in fields:
        s, a = size_align(t)
        off = (off + a - 1) // a * a + s
        align = max(align, a)
    return (off + align - 1) // align * align


CTYPES_OF = {"rt2d_material": _ffi.Material2D, "rt2d_scene_desc": _ffi.Scene2DDesc, "rt2d_render_params": _ffi.RenderParams2D,
             "rt3d_scene_desc": _ffi.Scene3DDesc, "rt3d_render_params": _ffi.RenderParams3D, "rt_stats": _ffi.Stats,
             "rt2d_denoiser": _ffi.Denoiser2D, "rt2d_progress": _ffi.Progress2D, "rt2d_forward_params": _ffi.ForwardParams2D}


def test_repr_c_structs_match_the_header_and_the_ctypes_layouts():
    cs, _ = parse_header()
    rs, _ = parse_rust()
    opaque = {n for n, f in rs.items() if [x[0] for x in f] == ["_p"]}
    assert {"rt_ctx", "rt2d_scene", "rt3d_scene", "rt2d_progressive"} <= opaque
    assert sorted(cs) == sorted(set(rs) - opaque), "a struct is declared on one side only"
    for name, fields in cs.items():
        assert fields == rs[name], f"{name}: fields differ\n header {fields}\n rust   {rs[name]}"
        assert name in CTYPES_OF, f"{name} has no ctypes counterpart in the Python binding"
        ct = CTYPES_OF[name]
        assert [f[0] for f in ct._fields_] == [f[0] for f in fields], f"{name}: ctypes field order differs"
        assert ctypes.sizeof(ct) == c_layout_size(fields, cs), f"{name}: ctypes size {ctypes.sizeof(ct)}"


def test_the_parsers_see_a_renamed_or_reordered_parameter():
    """Guards the guard: the comparison above must be sensitive to what it claims to catch."""
    _, cf = parse_header()
    ret, params = cf["rt2d_trace_batch"]
    assert ("ox", "*const f64") in params and params[0] == ("ctx", "*mut rt_ctx") and ret == "i32"
    assert cf["rt_version"] == ("*const c_char", [])
    assert cf["rt2d_progressive_planes"][1][1] == ("d_sum", "*mut *mut c_void")
    cs, _ = parse_header()
    assert ("cam_pos", "[f64; 3]") in cs["rt3d_scene_desc"] and ("seg_pid", "*const i32") in cs["rt2d_scene_desc"]
