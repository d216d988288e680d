"""The shared library is tied to the sources it was built from (round 5): build.py bakes a hash of every HIP source (with its
flags) and header into the library (`nic_build_id()`), the ctypes loader compares it with the hash of the files next to the
library and refuses a stale binary at load time.  CPU-only: symbols, no compute."""
import os
import shutil

import pytest

from neural_inventory_control_amd import _lib, build

pytestmark = pytest.mark.skipif(not _lib.library_built(), reason="libnic_hip.so not built (python -m neural_inventory_control_amd.build)")


def test_library_carries_the_id_of_the_sources_next_to_it():
    lib = _lib.load_library()
    bid = lib.nic_build_id().decode()
    assert len(bid) == 16 and bid == build.source_id()
    assert _lib.check_build_id(lib, build.CSRC) == bid


def test_an_edited_header_copy_is_a_mismatch_at_load_time(tmp_path):
    # the same tree shape as the package: <tmp>/pkg/csrc + <tmp>/include (HEADERS names the public header relative to csrc)
    csrc = tmp_path / "pkg" / "csrc"
    csrc.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    for src, _ in build.SOURCES:
        shutil.copy(os.path.join(build.CSRC, src), csrc / src)
    for h in build.HEADERS:
        shutil.copy(os.path.join(build.CSRC, h), os.path.normpath(os.path.join(csrc, h)))
    lib = _lib.load_library()
    assert build.source_id(str(csrc)) == lib.nic_build_id().decode()       # an untouched copy: same id (content, not mtime)
    os.utime(csrc / "nic_common.h")                                          # touching without changing: still the same
    assert _lib.check_build_id(lib, str(csrc)) == lib.nic_build_id().decode()
    with open(csrc / "nic_common.h", "a") as f:
        f.write("\n// edited after the build\n")
    assert build.source_id(str(csrc)) != lib.nic_build_id().decode()
    with pytest.raises(_lib.NicError, match="built from other sources"):
        _lib.check_build_id(lib, str(csrc))
    # a directory without the sources (an installed binary): nothing to compare, no error
    assert _lib.check_build_id(lib, str(tmp_path / "include")) is None
