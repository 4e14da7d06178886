"""CPU-side checks of the C-ABI library: it loads and exports every symbol the header declares
(no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "facl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"(?:int|int64_t)\s+(facl_\w+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    from facl_amd import _lib, build
    build.build()
    lib = _lib.load_library()
    syms = _header_symbols()
    assert len(syms) >= 5
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/facl_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in facl_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"
    assert lib.facl_version() >> 16 == 1


def test_header_constants_match_the_python_binding():
    """Buffer-size constants the ABI shares with callers (include/facl_hip.h) vs facl_amd/_lib.py."""
    from facl_amd import _lib
    txt = open(os.path.join(ROOT, "include", "facl_hip.h")).read()
    m = re.search(r"#define\s+FACL_AMAX_WORDS\s+(\d+)", txt)
    assert m and int(m.group(1)) == _lib.AMAX_WORDS
    common = open(os.path.join(ROOT, "facl_amd", "csrc", "common.h")).read()
    slots = int(re.search(r"#define\s+FACL_AMAX_SLOTS\s+(\d+)", common).group(1))
    stride = int(re.search(r"#define\s+FACL_AMAX_STRIDE\s+(\d+)", common).group(1))
    assert slots * stride == _lib.AMAX_WORDS and stride * 4 == 128           # one 128-byte line per slot


def test_no_cpu_fallback():
    """Product ops refuse CPU tensors instead of silently computing somewhere else."""
    import torch
    from facl_amd import utils_my
    with pytest.raises(RuntimeError):
        utils_my.knn_radius_group(torch.zeros(1, 64, 4), 8, 8, 0.1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "facl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
