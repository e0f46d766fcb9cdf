"""CPU-side checks of the drop-in boundary: libsvo_hip.so loads and exports every symbol that
include/svo_hip.h declares; no compute is attempted without a GPU; the product path never
falls back to the oracle."""
import ctypes as C
import os
import re

import pytest

from android_svo_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "svo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svo_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hip.load_library()
    names = declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert b"gfx950" in lib.svo_hip_version()


def test_struct_layouts_match_header():
    # sizes the C side was compiled with (natural alignment, LP64)
    assert C.sizeof(hip.CCamera) == 2 * 4 + 4 * 8 + 5 * 8 + 8          # int pad included
    assert C.sizeof(hip.CSiaParams) == 32
    assert C.sizeof(hip.CSiaResult) == 7 * 8 + 8 + 36 * 8 + 8 + 4 + 8 * 4 + 4 + 16
    assert C.sizeof(hip.CDfParams) == 24
    assert C.sizeof(hip.CSeedEvent) == 56 == hip.SEED_EVENT_DTYPE.itemsize       # svo_hip_seed_event: 2 x i32, 2 x f32, 5 x f64
    assert C.sizeof(hip.CTrackerConfig) == 96 and C.sizeof(hip.CTrackerMap) == 160
    assert C.sizeof(hip.CTrackResult) == 14 * 8 + 8 + 8 * 4 + 2 * 4 + 2 * 8 + 2 * 4 + 32 * 4 + 2 * 4 + C.sizeof(hip.CPoseOptResult)


def test_no_gpu_means_loud_failure_not_fallback():
    lib = hip.load_library()
    n = C.c_int(-1)
    rc = lib.svo_hip_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hip.SvoHipError):
        hip.Context(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "android_svo_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.replace("oracle/", "oracle/") or fn == "__init__.py" or \
                    not re.search(r"^\s*(from|import)\s+oracle|#include\s+\".*oracle", src, flags=re.M), fn
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
                assert "svo_oracle" not in src, fn


def test_missing_librccl_is_an_error_code_not_a_crash(tmp_path):
    """A deployment without a loadable librccl (the single-GPU case the header promises "needs no RCCL at all"):
    svo_hip_comm_unique_id must return SVO_HIP_ERR_DEVICE.  The library is hidden from a child process by a preloaded
    dlopen that fails for every name containing "rccl"; the child calls the entry point twice (the failure is cached)."""
    import subprocess
    import sys
    shim_c = tmp_path / "hide_rccl.c"
    shim_c.write_text(
        '#define _GNU_SOURCE\n#include <dlfcn.h>\n#include <string.h>\n'
        'void* dlopen(const char* name, int flags) {\n'
        '  static void* (*real)(const char*, int);\n'
        '  if (!real) real = (void* (*)(const char*, int))dlsym(RTLD_NEXT, "dlopen");\n'
        '  if (name && strstr(name, "rccl")) return 0;   /* and no dlerror() text: the worst case for the caller */\n'
        '  return real(name, flags);\n}\n')
    shim = tmp_path / "hide_rccl.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", str(shim_c), "-o", str(shim), "-ldl"])
    code = ("import ctypes as C\n"
            "lib = C.CDLL(%r)\n"
            "buf = C.create_string_buffer(128)\n"
            "print(lib.svo_hip_comm_unique_id(buf), lib.svo_hip_comm_unique_id(buf))\n" % hip.LIB_PATH)
    env = dict(os.environ, LD_PRELOAD=str(shim))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    assert r.stdout.split() == ["-2", "-2"], r.stdout
