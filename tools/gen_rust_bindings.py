#!/usr/bin/env python
"""Generates bindings/rust/imagepipe_amd_sys.rs -- the `extern "C"` block and `#[repr(C)]` structs a maintainer of the
reference crate (Rust) adds to call libimagepipe_amd.so -- from include/imagepipe_amd.h.

No rustc exists in this image, so the file cannot be compiled here; tests/test_rust_binding.py instead parses BOTH texts
independently and checks every struct field (name, order, type, offset against a gcc-compiled offsetof table) and every
function (name, argument count, argument and return types), so a header change without a binding change fails a test.

usage: python tools/gen_rust_bindings.py [--check]     (--check: exit 1 when the committed file is stale)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "imagepipe_amd.h")
OUT = os.path.join(ROOT, "bindings", "rust", "imagepipe_amd_sys.rs")

SCALARS = {"int": "c_int", "size_t": "usize", "float": "f32", "double": "f64", "char": "c_char", "uint8_t": "u8", "uint16_t": "u16",
           "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "int32_t": "i32", "void": "c_void", "unsigned": "c_uint"}
RUST_KEYWORDS = {"type", "ref", "in", "fn", "loop", "match", "move", "box", "try"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def camel(name):                       # ipk_pipeline_desc -> IpkPipelineDesc
    return "".join(p.capitalize() for p in name.split("_"))


def parse_header(text=None):
    """-> (enums, structs, opaque, functions); structs: [(name, [(ctype, field, array_len|None)])];
    functions: [(name, ret_ctype, [(ctype, argname)])].  ctype strings are normalised ('const float *', 'size_t')."""
    text = strip_comments(text if text is not None else open(HEADER).read())
    enums = []
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        items = []
        for it in m.group(1).split(","):
            it = it.strip()
            if it:
                k, v = [s.strip() for s in it.split("=")]
                items.append((k, int(v)))
        enums.append((m.group(2), items))
    structs = []
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ctype, rest = decl.split(" ", 1)
            for f in rest.split(","):
                f = f.strip()
                am = re.match(r"(\w+)\[(\d+)\]$", f)
                fields.append((ctype, am.group(1), int(am.group(2))) if am else (ctype, f, None))
        structs.append((m.group(2), fields))
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text)
    defines = [(m.group(1), int(m.group(2))) for m in re.finditer(r"#define\s+(IPK_\w+)\s+(\d+)\s*$", text, flags=re.M)]
    fnptrs = []
    for m in re.finditer(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\(([^)]*)\)\s*;", text):
        args = []
        for a in " ".join(m.group(3).split()).split(","):
            a = " ".join(a.replace("*", " * ").split())
            am = re.match(r"(.*?)(\w+)$", a)
            args.append((am.group(1).strip(), am.group(2)))
        fnptrs.append((m.group(2), m.group(1), args))
    funcs = []
    for m in re.finditer(r"IPK_API\s+([\w\s\*]+?)\b(ipk_\w+)\s*\(([^)]*)\)\s*;", text):
        ret = " ".join(m.group(1).replace("*", " * ").split())
        args = []
        argtext = " ".join(m.group(3).split())
        if argtext and argtext != "void":
            for a in argtext.split(","):
                a = " ".join(a.replace("*", " * ").split())
                am = re.match(r"(.*?)(\w+)$", a)
                args.append((am.group(1).strip(), am.group(2)))
        funcs.append((m.group(2), ret, args))
    parse_header.defines, parse_header.fnptrs = defines, fnptrs
    return enums, structs, opaque, funcs


def rust_type(ctype, struct_names, opaque):
    """C type text -> Rust type text"""
    toks = ctype.split()
    stars = toks.count("*")
    toks = [t for t in toks if t != "*"]
    const = "const" in toks
    base = [t for t in toks if t != "const"]
    assert len(base) == 1, ctype
    b = base[0]
    r = SCALARS.get(b) or (camel(b) if (b in struct_names or b in opaque or b in [f[0] for f in getattr(parse_header, "fnptrs", [])]) else None)
    assert r, "unmapped C type: " + ctype
    if stars == 0:
        return r
    # `const T *` -> *const T; `T *` -> *mut T; `const void *const *` -> *const *const c_void; `void *const *` -> *const *mut c_void
    # the header writes pointer-to-pointer arguments as `<inner> *const *name`: the const between the stars applies to the outer pointer
    if stars == 1:
        return ("*const " if const else "*mut ") + r
    m = re.match(r"^(const\s+)?(\w+)\s*\*\s*(const\s*)?\*$", ctype)
    assert m, ctype
    inner = ("*const " if m.group(1) else "*mut ") + r
    return ("*const " if m.group(3) else "*mut ") + inner


def generate():
    enums, structs, opaque, funcs = parse_header()
    sn = [s[0] for s in structs]
    o = []
    o.append("// imagepipe_amd_sys.rs -- raw FFI declarations for libimagepipe_amd.so (include/imagepipe_amd.h).")
    o.append("// GENERATED by tools/gen_rust_bindings.py from the header; checked field by field and symbol by symbol by")
    o.append("// tests/test_rust_binding.py (this image has no rustc, so the file is parsed, not compiled).")
    o.append("// Add to the reference crate as `mod imagepipe_amd_sys;` and link with `-limagepipe_amd`.")
    o.append("#![allow(non_camel_case_types, dead_code)]")
    o.append("use std::os::raw::{c_char, c_int, c_uint, c_void};")
    o.append("")
    for name, items in enums:
        o.append("// %s" % name)
        for k, v in items:
            o.append("pub const %s: c_int = %d;" % (k, v))
        o.append("")
    for k, v in parse_header.defines:
        o.append("pub const %s: usize = %d;" % (k, v))
    o.append("")
    for name in opaque:
        o.append("#[repr(C)] pub struct %s { _private: [u8; 0] }   // opaque: %s" % (camel(name), name))
    o.append("")
    for name, ret, args in parse_header.fnptrs:
        al = ", ".join("%s: %s" % (a, rust_type(ty, sn, opaque)) for ty, a in args)
        o.append("pub type %s = Option<unsafe extern \"C\" fn(%s) -> %s>;   // %s" % (camel(name), al, rust_type(ret, sn, opaque), name))
    o.append("")
    for name, fields in structs:
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy)]")
        o.append("pub struct %s {   // %s" % (camel(name), name))
        for ctype, f, n in fields:
            rt = rust_type(ctype, sn, opaque)
            o.append("  pub %s: %s," % (f, "[%s; %d]" % (rt, n) if n is not None else rt))
        o.append("}")
        o.append("")
    o.append('#[link(name = "imagepipe_amd")]')
    o.append('extern "C" {')
    for name, ret, args in funcs:
        al = ", ".join("%s: %s" % (("r#" + a) if a in RUST_KEYWORDS else a, rust_type(t, sn, opaque)) for t, a in args)
        rr = "" if ret == "void" else " -> " + rust_type(ret, sn, opaque)
        o.append("  pub fn %s(%s)%s;" % (name, al, rr))
    o.append("}")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print("wrote", OUT)
