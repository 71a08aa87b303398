"""The C ABI's promise "every function returns 0 or a negative RVO_E_* code; nothing throws or calls exit() across the ABI"
(include/rvo.h), probed with ~90 bad calls in a process of its own (tests/helpers/abi_error_probe.py): a segfault there is a
failed test here, with the call it died in."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bad_arguments_come_back_as_error_codes():
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "abi_error_probe.py")
    p = subprocess.run([sys.executable, probe], capture_output=True, text=True, timeout=110)
    tail = "\n".join(p.stdout.strip().splitlines()[-6:])
    assert p.returncode == 0, f"exit {p.returncode}\n{tail}\n{p.stderr[-600:]}"
    assert "PROBE DONE bad = 0 ok = True" in p.stdout
