""""The profiles describe this tree" as a mechanical check (a refactor changed the instruction stream of a
default-path kernel after its last GPU run and nobody noticed).  tools/isa_stamp.py hashes the gfx950 instruction stream of every
kernel of every csrc/*.hip (hipcc -S with the Makefile's flags); profiles/r06_ISA.json is the stamp committed with the tree and
profiles/r05_ISA.json the stamp of dce9137, the commit the round-5 profiles and the last green GPU suite were taken on.
CPU-only (hipcc cross-compiles without a GPU); about a minute."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")


@pytest.fixture(scope="module")
def tree_stamp():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "isa_stamp.py")], cwd=ROOT, timeout=900)
    return json.loads(out)["files"]


def test_committed_stamp_is_the_stamp_of_this_tree(tree_stamp):
    committed = json.load(open(os.path.join(ROOT, "profiles", "r06_ISA.json")))["files"]
    assert committed == tree_stamp, "csrc/*.hip changed since profiles/r06_ISA.json was written: re-run tools/isa_stamp.py --out profiles/r06_ISA.json " \
                                    "(and re-collect the profiles that describe the changed kernels)"


def test_every_kernel_measured_in_round_5_is_instruction_identical(tree_stamp):
    """The product library's kernels are the ones the round-5 profiles (profiles/r05_*) and the last builder-run GPU suite measured:
    none changed, none removed.  Kernels may be ADDED only in files that commit did not have (the red-zone scan, a debugging aid)."""
    old = json.load(open(os.path.join(ROOT, "profiles", "r05_ISA.json")))["files"]
    for f, ks in old.items():
        assert f in tree_stamp
        for k, h in ks.items():
            assert tree_stamp[f].get(k) == h, "%s: %s differs from the kernel measured at dce9137" % (f, k)
        if os.environ.get("STATTN_ALLOW_NEW_KERNELS") is None:
            assert set(tree_stamp[f]) == set(ks), "%s carries kernels that have never been measured: %s" % (f, sorted(set(tree_stamp[f]) - set(ks)))
    assert sorted(set(tree_stamp) - set(old)) == ["redzone.hip"]
