"""GPU: the reference's own checkers (test/check_from_file.hpp, test/check.hpp) re-expressed in C++ over
the header-only facade include/sshash_amd.hpp -- tests/cpp/check_dictionary.cpp, built by build()."""
from __future__ import annotations

import os
import subprocess

import pytest

from conftest import K63_FASTA, ROOT, SE_FASTA

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "tests", "cpp", "check_dictionary")


@pytest.mark.parametrize("fasta,k,m,extra", [(SE_FASTA, 31, 13, []), (SE_FASTA, 31, 17, ["--canonical"]),
                                             (K63_FASTA, 63, 25, []), (K63_FASTA, 63, 21, ["--canonical"])])
def test_reference_style_checkers_cpp(fasta, k, m, extra):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sshash_amd", "csrc"), "tools"])
    p = subprocess.run([BIN, fasta, str(k), str(m)] + extra, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "EVERYTHING OK!" in p.stdout
