"""
The Rust shim (rust/forge-ec-gpu/src/lib.rs) cannot be compiled in the build image (no rustc), so its
FFI surface is checked mechanically instead: every prototype of include/fecgpu.h and every declaration
of the shim's `extern "C"` block are parsed and compared -- same symbols, same arity, same argument and
return types (C type -> the Rust FFI type a binding must use for it).  Also: no unimplemented!()/todo!()
in the crate, all three curves implement GpuCurve, and the additive patch only adds lines.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fecgpu.h")
SHIM = os.path.join(ROOT, "rust", "forge-ec-gpu", "src", "lib.rs")
PATCH = os.path.join(ROOT, "rust", "forge-ec-curves-raw-coords.patch")

ENUMS = {"fec_curve", "fec_field_opcode", "fec_point_opcode", "fec_status"}  # C enums cross the ABI as int


def c_to_rust(t):
    t = re.sub(r"\s+", " ", t.strip())
    parts = t.split("*")          # "const uint64_t* const*" -> ["const uint64_t", " const", ""]
    stars = len(parts) - 1
    base = parts[0].replace("const", "").strip()
    prim = {"int": "c_int", "unsigned": "c_uint", "size_t": "usize", "uint64_t": "u64", "uint8_t": "u8", "void": "c_void", "char": "c_char",
            "float": "c_float", "double": "c_double", "fec_ctx": "FecCtx"}
    if base in ENUMS:
        base = "int"
    r = prim[base]
    if stars == 0:
        return "()" if r == "c_void" else r
    # every pointer level is *const when what it points to is const-qualified in the C declaration (the qualifier in
    # front of the base type for the innermost level, the one written after the previous `*` for the outer ones)
    out = r
    for level in range(stars):
        out = ("*const " if "const" in parts[level] else "*mut ") + out
    return out


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b((?:const\s+)?[A-Za-z_0-9]+\s*\**)\s*(fec_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            if a == "void":
                continue
            mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", a)  # type + parameter name
            params.append(c_to_rust(mm.group(1)))
        protos[name] = (params, c_to_rust(ret))
    return protos


def parse_shim():
    src = open(SHIM).read()
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S).group(1)
    decls = {}
    for m in re.finditer(r"fn\s+(fec_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        params = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in args.split(",") if ":" in a]  # r#in: raw identifier
        decls[name] = (params, re.sub(r"\s+", " ", ret))
    return decls


def test_every_header_symbol_is_declared_identically_in_the_shim():
    h, r = parse_header(), parse_shim()
    assert len(h) >= 25, sorted(h)
    assert set(h) == set(r), "only in header: %s; only in shim: %s" % (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    for name in sorted(h):
        assert h[name] == r[name], "%s:\n header -> %s\n shim   -> %s" % (name, h[name], r[name])


def test_header_symbols_match_the_ctypes_table():
    import sys
    sys.path.insert(0, ROOT)
    from forge_ec_amd import _lib
    assert set(parse_header()) == set(_lib.ABI_SYMBOLS)


def test_shim_is_complete():
    src = open(SHIM).read()
    assert "unimplemented!" not in src and "todo!" not in src
    for curve in ("secp256k1::Secp256k1", "p256::P256", "ed25519::Ed25519"):
        assert re.search(r"impl(_weierstrass!\(|\s+GpuCurve\s+for\s+)%s" % re.escape(curve), src), curve
    for api in ("batch_multiply", "batch_multiply_fixed", "batch_double_multiply", "batch_to_affine", "batch_compress",
                "multi_scalar_multiply", "ecdsa_verify_batch_secp256k1", "schnorr_batch_verify_secp256k1", "new_multi"):
        assert re.search(r"pub fn %s\b" % api, src), api
    # every extern function is actually used by the safe layer
    for name in parse_shim():
        assert len(re.findall(r"\b%s\(" % name, src)) >= 2, name + " declared but never called"


def test_patch_is_purely_additive():
    body = [l for l in open(PATCH).read().split("\n") if not l.startswith(("---", "+++", "diff ", "@@"))]
    assert not [l for l in body if l.startswith("-")], "the patch removes or changes reference lines"
    added = [l for l in body if l.startswith("+")]
    assert len(added) > 60 and sum("from_raw_coords" in l for l in added) >= 6
