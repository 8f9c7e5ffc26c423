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
