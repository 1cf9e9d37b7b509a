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
    assert len(re.findall(r"^\s*from oracle|^\s*import oracle", bench, flags=re.M)) == 1   # inside cpu_baseline()
    assert "def cpu_baseline" in bench and bench.index("def cpu_baseline") < bench.index("from oracle")
