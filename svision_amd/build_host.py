"""Compiles the per-read host modules of the collection step with Cython, in place (``python -m svision_amd.build_host``).

The reference's collection is interpreted Python and so is this package's mirror of it (SURVEY 8(a'): same functions, same
order of operations).  The six modules below are where a 10 Mb window spends its ~90 ms of host time; compiled as they
are -- pure-Python mode, no type annotations, no semantic change -- they take 1.5x less (tools/exp/prof_collect.py); since round 3 the hot ones
carry static types in augmenting ``.pxd`` files next to them (segments and coordinates as C structs / longs: another 2x,
same source, same results), and an extension module next to its ``.py`` source is what ``import`` picks up.  Without the build the ``.py`` files run: same
results, slower.  Outputs (``*.so``, generated ``*.c``) are git-ignored; the ``.so`` files travel with the snapshot like
``libsvx.so``."""
import os
import sys

MODULES = ["collection/analyze_reads.py", "collection/collect_signatures.py", "collection/classes.py",
           "collection/output_clusters.py", "collection/cluster_signatures.py", "collection/graph.py",
           "segmentplot/classes.py", "network/predict.py", "network/genotype.py"]


STAMP = "_host_build.json"               # per module: sha1 of the sources its extension module was compiled from (_source_hash)


def _source_hash(here, m):
    """sha1 of everything the extension module of ``m`` was compiled from: its ``.py`` and the ``.pxd`` files of ALL the
    modules (they declare the extension types' C layout, which the modules share: a changed ``classes.pxd`` makes every
    binary stale, not only its own).  Without any ``.pxd`` this is the sha1 of the ``.py``."""
    import hashlib
    h = hashlib.sha1()
    with open(os.path.join(here, m), "rb") as f:
        h.update(f.read())
    for other in MODULES:
        pxd = os.path.join(here, other[:-3] + ".pxd")
        if os.path.exists(pxd):
            h.update(other.encode())
            with open(pxd, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def compiled_state(package="svision_amd"):
    """-> (modules running compiled, modules running interpreted): which form of the host modules this process imports.
    The command line logs it and bench.py reports it: without the build the same Python runs 2-4x slower, silently."""
    import importlib
    compiled, interpreted = [], []
    for m in MODULES:
        mod = importlib.import_module(package + "." + m[:-3].replace("/", "."))
        (compiled if getattr(mod, "__file__", "").endswith(".so") else interpreted).append(m[:-3])
    return compiled, interpreted


def stale_modules(here=None):
    """-> [(module path relative to the package, [its extension files])] for every compiled host module whose ``.py``
    source is not the one it was compiled from.  Read-only (eight small files are hashed)."""
    import json
    here = here or os.path.dirname(os.path.abspath(__file__))
    try:
        with open(os.path.join(here, STAMP)) as f:
            stamp = json.load(f)
    except (OSError, ValueError):
        stamp = {}
    stale = []
    for m in MODULES:
        src = os.path.join(here, m)
        d, base = os.path.dirname(src), os.path.basename(m)[:-3]
        try:
            sos = [n for n in os.listdir(d) if n.startswith(base + ".") and n.endswith(".so")]
        except OSError:
            continue
        if sos and stamp.get(m) != _source_hash(here, m):
            stale.append((m, sos))
    return stale


class _SourceFirst:
    """Meta-path finder: the listed modules are imported from their ``.py`` source although an extension module of the
    same name sits next to it.  What ``import svision_amd`` installs when it finds stale binaries: nothing on disk is
    touched (no race between ranks, works on a read-only install), and the stale code cannot run."""

    def __init__(self, sources):
        self.sources = dict(sources)                          # {fully qualified module name: path of its .py}

    def find_spec(self, fullname, path=None, target=None):
        py = self.sources.get(fullname)
        if py is None:
            return None
        import importlib.util
        return importlib.util.spec_from_file_location(fullname, py)


def guard_imports(package="svision_amd", here=None, log=None):
    """Called on package import: stale extension modules are bypassed (their sources run interpreted) and reported.
    ``SVX_HOST_INTERPRETED=1`` bypasses ALL of them: the same sources as plain Python (tests/test_host_typed.py runs the
    golden collection and vote that way -- the form a machine without the build executes)."""
    here = here or os.path.dirname(os.path.abspath(__file__))
    if os.environ.get("SVX_HOST_INTERPRETED"):
        names = {package + "." + m[:-3].replace("/", "."): os.path.join(here, m) for m in MODULES}
        sys.meta_path.insert(0, _SourceFirst(names))
        return sorted(names)
    stale = stale_modules(here)
    if not stale:
        return []
    # ONE stale module bypasses ALL of them: the typed modules cimport each other's extension types (Seg, Segment), so a
    # compiled collect_signatures next to an interpreted classes dies at import (KeyError '__pyx_vtable__') -- a mixed
    # state is not a state this package can run in.
    names = {package + "." + m[:-3].replace("/", "."): os.path.join(here, m) for m in MODULES}
    sys.meta_path.insert(0, _SourceFirst(names))
    if log is not None:
        log("svision_amd: compiled host modules older than their source (" + ", ".join(m for m, _s in stale) + "): ALL host "
            "modules run interpreted from their .py files (rebuild with `python -m svision_amd.build_host`)")
    return sorted(names)


def drop_stale(log=None, here=None):
    """Remove extension modules whose ``.py`` source changed since they were compiled.  Build-time housekeeping
    (``build()`` below, ``clean``); importing the package never deletes anything (see :func:`guard_imports`)."""
    here = here or os.path.dirname(os.path.abspath(__file__))
    dropped = []
    for m, sos in stale_modules(here):
        d = os.path.dirname(os.path.join(here, m))
        for n in sos:
            try:
                os.remove(os.path.join(d, n))
                dropped.append(os.path.join(os.path.dirname(m), n))
            except FileNotFoundError:
                pass
            except OSError as exc:
                if log is not None:
                    log("svision_amd: cannot remove stale %s: %s" % (os.path.join(d, n), exc))
    if dropped and log is not None:
        log("svision_amd: removed compiled host modules older than their source (run `python -m svision_amd.build_host`): " + ", ".join(dropped))
    return dropped


def build(quiet=True):
    import json
    import shutil
    import tempfile
    from setuptools import setup
    from Cython.Build import cythonize
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    cwd = os.getcwd()
    os.chdir(root)
    drop_stale(log=lambda msg: print(msg, file=sys.stderr))
    tmp = tempfile.mkdtemp(prefix="svx_host_build_")
    os.environ["CFLAGS"] = (os.environ.get("CFLAGS", "") + " -g0 -O2").strip()           # no debug info: 8 x 0.1 MB instead of 8 x 1 MB
    try:
        ext = cythonize([os.path.join("svision_amd", m) for m in MODULES], language_level=3, quiet=quiet, build_dir=tmp,
                        compiler_directives={"binding": True, "embedsignature": True})
        setup(name="svision_amd_host", ext_modules=ext,
              script_args=["build_ext", "--inplace", "--build-temp", tmp, "--build-lib", tmp] + (["-q"] if quiet else []))
        with open(os.path.join(here, STAMP), "w") as f:
            json.dump({m: _source_hash(here, m) for m in MODULES}, f, indent=0, sort_keys=True)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def clean():
    here = os.path.dirname(os.path.abspath(__file__))
    for m in MODULES:
        base = os.path.join(here, m[:-3])
        d = os.path.dirname(base)
        for name in os.listdir(d):
            if name.startswith(os.path.basename(base) + ".") and (name.endswith(".so") or name.endswith(".c")):
                os.remove(os.path.join(d, name))


if __name__ == "__main__":
    clean() if sys.argv[1:] == ["clean"] else build(quiet="-v" not in sys.argv)
