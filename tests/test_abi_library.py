"""CPU tests of the drop-in boundary: libfecgpu.so builds for gfx950, loads, exports every symbol
include/fecgpu.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from forge_ec_amd import build
    build.build()
    from forge_ec_amd import _lib
    return _lib.lib()


def _declared_symbols(header="fecgpu.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fec_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree(lib):
    from forge_ec_amd import _lib
    declared = _declared_symbols()
    assert declared == sorted(_lib.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    canon = _declared_symbols("fecgpu_canon.h")
    assert canon == sorted(_lib.CANON_ABI_SYMBOLS)
    for s in canon:
        assert hasattr(lib, s), s


def test_point_limbs_and_strerror(lib):
    assert lib.fec_point_limbs(0) == 12 and lib.fec_point_limbs(1) == 12 and lib.fec_point_limbs(2) == 16
    assert lib.fec_point_limbs(9) == 0
    assert lib.fec_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.fec_strerror(-2)


def test_no_cpu_fallback_without_gpu(lib):
    """On a box without a gfx950 GPU, ctx creation fails loudly instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert lib.fec_ctx_create(ctypes.byref(h), 0) == -2
    assert not h.value
    import forge_ec_amd as F
    with pytest.raises(F.FecError):
        F.Context(0)


def test_product_code_never_touches_the_oracle():
    """The shipped package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "forge_ec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("CPU oracle", "").replace("the oracle", "").replace(
                    "oracle-checked", "").replace("oracles", ""), os.path.join(dirpath, f)
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libfecgpu.so")], capture_output=True, text=True).stdout
    assert "forge_ec_oracle" not in out


def test_generated_field_asm_is_current(tmp_path):
    """forge_ec_amd/csrc/field_asm.inc is the output of tools/gen_field_asm.py with its default register blocks
    (all three curves below v168: every multiplication kernel runs three wavefronts per SIMD)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_field_asm", os.path.join(root, "tools", "gen_field_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.OUT = str(tmp_path / "field_asm.inc")
    gen.main()
    assert open(gen.OUT).read() == open(os.path.join(root, "forge_ec_amd", "csrc", "field_asm.inc")).read()
    assert gen.SECP_TOP == gen.P256_TOP == gen.ED_TOP == 168


def test_only_the_allowed_places_call_the_oracle():
    """oracle/ is test infrastructure: besides tests/, only __graft_entry__.smoke() and bench.py's cpu_baseline leg
    may import it -- no script under tools/, rust/, examples/ or the package does."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.+oracle\b)|libforge_ec_oracle", re.M)
    for top in ("tools", "forge_ec_amd", "rust", "examples", "include"):
        for dirpath, _, files in os.walk(os.path.join(root, top)):
            for f in files:
                if f.endswith((".py", ".sh", ".cpp", ".hip", ".hpp", ".h", ".c", ".rs")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert not pat.search(text), os.path.join(dirpath, f)
    bench = open(os.path.join(root, "bench.py")).read()
    assert len(re.findall(r"^\s*from oracle|^\s*import oracle", bench, flags=re.M)) == 1   # load_checker(): the cpu_baseline leg and its pre-build
    assert "def cpu_baseline" in bench and bench.index("def cpu_baseline") < bench.index("from oracle")


def _extern_c_definitions(path):
    """(name, text-after-signature) of every function DEFINED at namespace level inside the extern "C" block."""
    text = open(path).read()
    body = text[text.index('extern "C" {'):text.index('}  // extern "C"')]
    out = []
    for m in re.finditer(r"^(?:int|void|const char\*|const uint64_t\*) (fec_\w+)\(", body, flags=re.M):
        rest = body[m.end():]
        depth, i = 1, 0
        while depth:  # end of the parameter list
            depth += {"(": 1, ")": -1}.get(rest[i], 0)
            i += 1
        out.append((m.group(1), rest[i:i + 200]))
    return out


def test_no_exception_can_cross_the_c_abi():
    """Every extern "C" definition is a function-try-block closed by an FEC_ABI_CATCH_* macro (one-line bodies that
    touch nothing that can throw are listed), and a shard worker thread catches inside the thread."""
    trivial = {"fec_point_limbs", "fec_ctx_device_count"}
    seen = 0
    for src in ("fecgpu.hip", "canon.hip"):
        path = os.path.join(ROOT, "forge_ec_amd", "csrc", src)
        for name, after in _extern_c_definitions(path):
            seen += 1
            if name in trivial:
                continue
            assert after.lstrip().startswith("try {"), "%s in %s is not a function-try-block" % (name, src)
        text = open(path).read()
        assert text.count(") try {") == text.count("} FEC_ABI_CATCH_")
    from forge_ec_amd import _lib
    assert seen == len(_lib.ABI_SYMBOLS) + len(_lib.CANON_ABI_SYMBOLS)
    fec = open(os.path.join(ROOT, "forge_ec_amd", "csrc", "fecgpu.hip")).read()
    worker = fec[fec.index("workers[g] = std::thread("):]
    assert worker.index("try {") < worker.index("rc[g] = call(g)") < worker.index("catch (...)")


def test_ed25519_sort_header_layout():
    """The popcount sort's work-area header: hist, cursor and the permutation do not overlap (round 2 had
    cursor[252..256] on top of perm[0..4]); the layout is also a static_assert in the source."""
    text = open(os.path.join(ROOT, "forge_ec_amd", "csrc", "kernels_ed.hip")).read()
    bins = int(re.search(r"constexpr int ED_BINS = (\d+);", text).group(1))
    cursor_at = int(re.search(r"constexpr int ED_CURSOR_AT = (\d+);", text).group(1))
    perm_off = int(re.search(r"constexpr size_t ED_PERM_OFFSET = (\d+);", text).group(1))
    assert bins == 257 and cursor_at >= bins and (cursor_at + bins) * 4 <= perm_off
    assert "hist + ED_CURSOR_AT" in text and "hipMemsetAsync(hist, 0, ED_PERM_OFFSET" in text
    assert "static_assert(ED_CURSOR_AT >= ED_BINS" in text


def test_destroy_wipes_before_it_frees():
    fec = open(os.path.join(ROOT, "forge_ec_amd", "csrc", "fecgpu.hip")).read()
    d = fec[fec.index("void fec_ctx_destroy("):]
    d = d[:d.index("FEC_ABI_CATCH_VOID")]
    assert d.index("fec_ctx_wipe(ctx)") < d.index("hipFree("), "the wipe must come before the first hipFree"
