"""Repository contract checks (CPU): the C-ABI library loads and exports every symbol the header
declares; the product never imports the oracle; no CPU fallback."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "eftbird.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(eftb_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    from eftpipe_amd import _lib

    lib = _lib.load()
    declared = _header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"libeftbird.so lacks {name} declared in include/eftbird.h"
    assert sorted(_lib.EXPORTS) == declared, "eftpipe_amd/_lib.py EXPORTS out of sync with include/eftbird.h"


def test_enums_in_sync_with_header():
    from eftpipe_amd import _lib

    src = open(os.path.join(ROOT, "include", "eftbird.h")).read()
    tables = re.search(r"enum eftb_table \{(.*?)\};", src, re.S).group(1)
    names = re.findall(r"EFTB_T_([A-Z0-9]+)", re.sub(r"/\*.*?\*/", "", tables, flags=re.S))
    assert names[:-1] == _lib.TABLES and names[-1] == "COUNT"
    buffers = re.search(r"enum eftb_buffer \{(.*?)\};", src, re.S).group(1)
    names = re.findall(r"EFTB_B_([A-Z0-9]+)", re.sub(r"/\*.*?\*/", "", buffers, flags=re.S))
    assert names[:-1] == _lib.BUFFERS and names[-1] == "COUNT"


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "eftpipe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{f} imports the oracle"
                assert "oracle/" not in text and "oracle." not in text.replace("oracle.py", ""), f"{f} references the oracle"


def test_no_gpu_means_loud_failure():
    """Without a HIP device the engine must raise, never compute on the CPU."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from eftpipe_amd.engine import Engine\n"
        "from eftpipe_amd.tables import EngineConfig\n"
        "from eftpipe_amd._lib import EftbError\n"
        "try:\n"
        "    Engine(EngineConfig(Nl=2))\n"
        "except EftbError as e:\n"
        "    print('RAISED', e)\n" % ROOT
    )
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "RAISED" in out.stdout and "no HIP device" in out.stdout, out.stdout + out.stderr
