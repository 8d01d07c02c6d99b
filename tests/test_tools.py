"""The measurement tools under tools/ and profiles/ are not run by the suite (they need a GPU and
minutes); at least they must parse."""
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "profiles", "*.py")))


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, ROOT) for p in SCRIPTS])
def test_script_parses(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "out.pyc"), doraise=True)


def test_shell_scripts_parse():
    import subprocess
    for sh in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))):
        r = subprocess.run(["bash", "-n", sh], capture_output=True, text=True)
        assert r.returncode == 0, "%s: %s" % (sh, r.stderr)
