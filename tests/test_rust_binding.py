"""The Rust FFI stub (bindings/rust/imagepipe_amd_sys.rs, quoted by INTEGRATION.md) cannot be compiled here (no rustc), so it is
machine-checked against the C header instead: every #[repr(C)] struct field by name, order, type, size and OFFSET (the C side
from a gcc-compiled offsetof table, the Rust side from repr(C) layout rules applied to the parsed Rust text), every extern
function by name, arity and argument/return types, every enum constant by value.  A field or symbol added to
include/imagepipe_amd.h without the binding fails here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = os.path.join(ROOT, "bindings", "rust", "imagepipe_amd_sys.rs")
HDR = os.path.join(ROOT, "include", "imagepipe_amd.h")

RUST_SIZES = {"c_int": 4, "c_uint": 4, "i32": 4, "u32": 4, "f32": 4, "usize": 8, "u64": 8, "i64": 8, "f64": 8, "u8": 1, "c_char": 1, "u16": 2}
C_TO_RUST = {"int": "c_int", "size_t": "usize", "float": "f32", "char": "c_char", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32",
             "uint64_t": "u64", "int64_t": "i64", "void": "c_void", "ipk_cache": "IpkCache", "ipk_ctx": "IpkCtx", "ipk_pipeline_desc": "IpkPipelineDesc",
             "ipk_fused_params": "IpkFusedParams", "ipk_comm": "IpkComm", "ipk_band": "IpkBand", "ipk_stage_time": "IpkStageTime", "ipk_exchange_fn": "IpkExchangeFn"}


def _rust_structs(text):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{\s*// (\w+)\n(.*?)\n\}", text, flags=re.S):
        fields = []
        for fm in re.finditer(r"pub (\w+): ([^,]+),", m.group(3)):
            ty = fm.group(2).strip()
            am = re.match(r"\[(\w+); (\d+)\]", ty)
            fields.append((fm.group(1), am.group(1), int(am.group(2))) if am else (fm.group(1), ty, None))
        out[m.group(2)] = (m.group(1), fields)
    return out


def _rust_layout(fields):
    """repr(C): each field at the next multiple of its alignment; size rounded up to the largest alignment"""
    off, maxa, table = 0, 1, []
    for name, ty, n in fields:
        sz = 8 if ty.startswith("*") else RUST_SIZES[ty]
        off = (off + sz - 1) // sz * sz
        table.append((name, off, sz * (n or 1)))
        off += sz * (n or 1)
        maxa = max(maxa, sz)
    return table, (off + maxa - 1) // maxa * maxa


def _c_structs():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_bindings as g
    return g.parse_header()


def _c_offsets(tmp_path, structs):
    """offsetof/sizeof of every field as gcc lays the header's structs out"""
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "imagepipe_amd.h"', "int main(void) {"]
    for name, fields in structs:
        for _, f, _ in fields:
            src.append('  printf("%s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));' % (name, f, name, f, name, f))
        src.append('  printf("%s . %%zu 0\\n", sizeof(%s));' % (name, name))
    src.append("  return 0; }")
    c = tmp_path / "offs.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "offs"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = {}
    for line in subprocess.check_output([str(exe)]).decode().split("\n"):
        if line:
            s, f, o, z = line.split()
            out.setdefault(s, []).append((f, int(o), int(z)))
    return out


def test_rust_structs_match_header_layout(tmp_path):
    _, structs, _, _ = _c_structs()
    assert {s[0] for s in structs} >= {"ipk_pipeline_desc", "ipk_fused_params"}
    rs = _rust_structs(open(RS).read())
    coff = _c_offsets(tmp_path, structs)
    for name, cfields in structs:
        assert name in rs, "bindings/rust has no #[repr(C)] struct for " + name
        _, rfields = rs[name]
        assert [f for f, _, _ in rfields] == [f for _, f, _ in cfields], name + ": field names/order differ from the header"
        for (ctype, f, n), (rf, rty, rn) in zip(cfields, rfields):
            assert C_TO_RUST[ctype] == rty and n == rn, "%s.%s: C `%s[%s]` vs Rust `%s[%s]`" % (name, f, ctype, n, rty, rn)
        table, size = _rust_layout(rfields)
        want = coff[name]
        assert table == want[:-1], name + ": repr(C) offsets differ from gcc's"
        assert size == want[-1][1], name + ": size differs"


def test_rust_structs_match_ctypes_tables():
    from imagepipe_amd import _lib
    rs = _rust_structs(open(RS).read())
    for cname, ct in (("ipk_pipeline_desc", _lib.PipelineDesc), ("ipk_fused_params", _lib.FusedParams)):
        _, rfields = rs[cname]
        table, size = _rust_layout(rfields)
        assert size == C.sizeof(ct)
        assert [(f, o) for f, o, _ in table] == [(f, getattr(ct, f).offset) for f, _ in ct._fields_]


def _map_ctype(t):
    toks = t.split()
    stars = toks.count("*")
    base = [x for x in toks if x not in ("*", "const")]
    assert len(base) == 1, t
    r = C_TO_RUST[base[0]]
    if stars == 0:
        return r
    if stars == 1:
        return ("*const " if "const" in toks else "*mut ") + r
    m = re.match(r"^(const\s+)?(\w+)\s*\*\s*(const\s*)?\*$", t)
    return ("*const " if m.group(3) else "*mut ") + ("*const " if m.group(1) else "*mut ") + r


def test_rust_extern_block_declares_every_symbol_with_matching_types():
    enums, _, _, funcs = _c_structs()
    text = open(RS).read()
    ext = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    rfuncs = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)( -> ([^;]+))?;", ext):
        args = [a.split(": ", 1)[1].strip() for a in m.group(2).split(", ") if a.strip()]
        rfuncs[m.group(1)] = (args, (m.group(4) or "").strip())
    assert sorted(rfuncs) == sorted(f[0] for f in funcs), "extern block and header declare different symbol sets"
    for name, ret, args in funcs:
        ra, rr = rfuncs[name]
        assert len(ra) == len(args), name + ": arity"
        for (ct, an), rt in zip(args, ra):
            assert _map_ctype(ct) == rt, "%s(%s): C `%s` vs Rust `%s`" % (name, an, ct, rt)
        assert ("" if ret == "void" else _map_ctype(ret)) == rr, name + ": return type"
    for _, items in enums:
        for k, v in items:
            assert re.search(r"pub const %s: c_int = %d;" % (k, v), text), "enum constant %s" % k
    # the callback type of the host transport: argument types in order
    hdr = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    m = re.search(r"typedef\s+int\s*\(\s*\*\s*ipk_exchange_fn\s*\)\s*\(([^)]*)\)", hdr)
    cargs = [" ".join(a.replace("*", " * ").split()).rsplit(" ", 1)[0] for a in m.group(1).split(",")]
    rm = re.search(r"pub type IpkExchangeFn = Option<unsafe extern \"C\" fn\((.*?)\) -> c_int>;", text)
    rargs = [a.split(": ", 1)[1] for a in rm.group(1).split(", ")]
    assert [_map_ctype(a) for a in cargs] == rargs
    assert "pub const IPK_COMM_ID_BYTES: usize = 128;" in text


def test_committed_binding_is_current_and_integration_md_quotes_it():
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"]) == 0, \
        "bindings/rust/imagepipe_amd_sys.rs is stale: run tools/gen_rust_bindings.py"
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "bindings/rust/imagepipe_amd_sys.rs" in md
    # the struct sketch INTEGRATION.md shows inline must carry every field of ipk_pipeline_desc, in order
    _, structs, _, _ = _c_structs()
    fields = [f for _, f, _ in dict(structs)["ipk_pipeline_desc"]]
    block = re.search(r"pub struct IpkPipelineDesc \{(.*?)\n\}", md, flags=re.S).group(1)
    assert re.findall(r"pub (\w+):", block) == fields
