"""The C-ABI shared library loads and exports every symbol include/icikt.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "icikt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(icikt_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_exported():
    from icikendalltau_amd import _lib
    if _lib.needs_build():
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/icikt.h but not exported"
    assert sorted(_lib.EXPORTS) == names
    assert L.icikt_version() == 400


def test_constants_match_header():
    from icikendalltau_amd import _lib
    src = open(os.path.join(ROOT, "include", "icikt.h")).read()
    defs = dict(re.findall(r"#define\s+(ICIKT_[A-Z0-9_]+)\s+\(?(-?\d+)u?\)?", src))
    assert int(defs["ICIKT_CNT_FIELDS"]) == len(_lib.CNT_FIELDS)
    assert int(defs["ICIKT_PERSPECTIVE_LOCAL"]) == _lib.PERSPECTIVE["local"]
    assert int(defs["ICIKT_PERSPECTIVE_GLOBAL"]) == _lib.PERSPECTIVE["global"]
    assert int(defs["ICIKT_ALT_TWO_SIDED"]) == _lib.ALTERNATIVE["two.sided"]
    assert int(defs["ICIKT_ALT_LESS"]) == _lib.ALTERNATIVE["less"]
    assert int(defs["ICIKT_ALT_GREATER"]) == _lib.ALTERNATIVE["greater"]
    assert int(defs["ICIKT_MAX_FEATURES"]) == _lib.MAX_FEATURES
    assert int(defs["ICIKT_MAX_FEATURES_WIDE"]) == _lib.MAX_FEATURES_WIDE
    assert int(defs["ICIKT_E_NO_DEVICE"]) == _lib.E_NO_DEVICE
    for i, f in enumerate(_lib.CNT_FIELDS):
        assert int(defs["ICIKT_CNT_" + f.upper()]) == i


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a HIP device the product path must fail loudly, never compute on the CPU."""
    from icikendalltau_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.IciktError, match="no usable HIP device"):
        _lib.Context(0)
    import numpy as np
    from icikendalltau_amd import ici_kt
    with pytest.raises(_lib.IciktError):
        ici_kt(np.arange(10.0), np.arange(10.0))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "icikendalltau_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".c", ".R")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in text.lower() or fn == "api.py" and "import oracle" not in text and "from oracle" not in text, fn


def test_library_never_page_locks_or_probes_caller_memory():
    """By construction: the library's sources call neither hipHostRegister / hipHostUnregister (a
    per-call registration of caller heap ranges was the common factor of the two GPU memory-access faults of rounds 2
    and 3) nor hipPointerGetAttributes (probing pageable pointers floods the runtime's error log).  Comments may name
    them; code may not."""
    csrc = os.path.join(ROOT, "icikendalltau_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        code = []
        for ln in open(os.path.join(csrc, fn), errors="ignore"):
            code.append(ln.split("//", 1)[0])
        text = re.sub(r"/\*.*?\*/", "", "".join(code), flags=re.S)
        for sym in ("hipHostRegister", "hipHostUnregister", "hipPointerGetAttributes", "hipHostGetDevicePointer"):
            assert sym not in text, f"{fn} calls {sym}"
    from icikendalltau_amd import _lib
    assert _lib.FLAG_HOST_PINNED == 8
    src = open(os.path.join(ROOT, "include", "icikt.h")).read()
    assert re.search(r"#define\s+ICIKT_FLAG_HOST_PINNED\s+8u", src)
