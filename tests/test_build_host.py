"""The compiled host modules must never shadow a newer source (svision_amd/build_host.py: drop_stale)."""
import hashlib
import json
import os

from svision_amd import build_host


def _layout(root, stamp_matches):
    for m in build_host.MODULES:
        d = os.path.join(root, os.path.dirname(m))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(root, m), "w") as f:
            f.write("x = 1\n")
        with open(os.path.join(d, os.path.basename(m)[:-3] + ".cpython-310-x86_64-linux-gnu.so"), "wb") as f:
            f.write(b"\x7fELF")
    sha = hashlib.sha1(b"x = 1\n").hexdigest()
    with open(os.path.join(root, build_host.STAMP), "w") as f:
        json.dump({m: (sha if stamp_matches else "0" * 40) for m in build_host.MODULES}, f)


def test_modules_matching_their_stamp_are_kept(tmp_path):
    _layout(str(tmp_path), True)
    assert build_host.drop_stale(here=str(tmp_path)) == []
    assert all(os.path.exists(os.path.join(tmp_path, m[:-3] + ".cpython-310-x86_64-linux-gnu.so")) for m in build_host.MODULES)


def test_a_changed_source_drops_its_extension_module_only(tmp_path):
    _layout(str(tmp_path), True)
    changed = build_host.MODULES[0]
    with open(os.path.join(tmp_path, changed), "a") as f:
        f.write("y = 2\n")
    messages = []
    dropped = build_host.drop_stale(log=messages.append, here=str(tmp_path))
    assert len(dropped) == 1 and os.path.basename(changed)[:-3] in dropped[0] and messages
    assert not os.path.exists(os.path.join(tmp_path, changed[:-3] + ".cpython-310-x86_64-linux-gnu.so"))
    assert os.path.exists(os.path.join(tmp_path, build_host.MODULES[1][:-3] + ".cpython-310-x86_64-linux-gnu.so"))


def test_without_a_stamp_every_extension_module_goes(tmp_path):
    _layout(str(tmp_path), True)
    os.remove(os.path.join(tmp_path, build_host.STAMP))
    assert len(build_host.drop_stale(here=str(tmp_path))) == len(build_host.MODULES)


def test_import_guard_runs_the_source_of_a_stale_module_and_touches_nothing(tmp_path):
    """`import svision_amd` with a stale extension module on disk: the .py runs, the file stays (no deletion at import)."""
    import importlib
    import sys
    pkg = tmp_path / "fakepkg"
    _layout(str(pkg), True)
    for d in {os.path.dirname(m) for m in build_host.MODULES}:
        open(os.path.join(pkg, d, "__init__.py"), "w").close()
    open(pkg / "__init__.py", "w").close()
    changed = build_host.MODULES[0]
    with open(os.path.join(pkg, changed), "a") as f:
        f.write("y = 2\n")
    def binaries():
        return sorted(n for n in os.listdir(os.path.join(pkg, os.path.dirname(changed))) if n.endswith(".so"))
    before = binaries()
    messages = []
    sys.path.insert(0, str(tmp_path))
    n_finders = len(sys.meta_path)
    try:
        names = build_host.guard_imports(package="fakepkg", here=str(pkg), log=messages.append)
        # one stale module bypasses all of them (they cimport each other's extension types: ADVICE r3)
        assert names == sorted("fakepkg." + m[:-3].replace("/", ".") for m in build_host.MODULES) and messages
        mod = importlib.import_module("fakepkg." + changed[:-3].replace("/", "."))   # the fake .so is not a loadable ELF: only the source can import
        assert mod.y == 2 and mod.__file__.endswith(".py")
        other = importlib.import_module("fakepkg." + build_host.MODULES[1][:-3].replace("/", "."))
        assert other.__file__.endswith(".py")
        assert binaries() == before and len(before) >= 1
    finally:
        sys.path.remove(str(tmp_path))
        del sys.meta_path[:len(sys.meta_path) - n_finders]
        for k in [k for k in sys.modules if k.startswith("fakepkg")]:
            del sys.modules[k]


def test_a_changed_pxd_makes_every_extension_module_stale(tmp_path):
    """The .pxd files declare the C layout of the extension types the modules share."""
    _layout(str(tmp_path), True)
    assert build_host.stale_modules(str(tmp_path)) == []
    with open(os.path.join(tmp_path, build_host.MODULES[2][:-3] + ".pxd"), "w") as f:
        f.write("cdef class X:\n    cdef public long a\n")
    assert len(build_host.stale_modules(str(tmp_path))) == len(build_host.MODULES)


def test_compiled_state_lists_every_host_module_once():
    compiled, interpreted = build_host.compiled_state()
    assert sorted(compiled + interpreted) == sorted(m[:-3] for m in build_host.MODULES)


def test_real_package_imports_with_a_stale_shared_module(tmp_path):
    """classes.py edited without a rebuild -> every host module must run from source (a compiled
    collect_signatures next to an interpreted classes dies with KeyError '__pyx_vtable__').  Run on a copy of the package."""
    import shutil
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(build_host.__file__))
    if not build_host.compiled_state()[0]:
        import pytest
        pytest.skip("host modules are not compiled here")
    dst = tmp_path / "svision_amd"
    shutil.copytree(here, dst, ignore=shutil.ignore_patterns("__pycache__", "csrc", "libsvx*.so"))
    with open(dst / "collection" / "classes.py", "a") as f:
        f.write("\n# edited after the build\n")
    code = ("import svision_amd.collection.collect_signatures as m, svision_amd.collection.classes as c; "
            "assert m.__file__.endswith('.py') and c.__file__.endswith('.py'); print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=str(tmp_path)))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
