"""Load the reference's hot-path modules (read-only, build container only).

TEST/FIXTURE INFRASTRUCTURE -- never imported by the product, never run on the GPU box.

The reference package cannot be imported whole (``eftpipe/__init__.py`` pulls in cobaya,
which is not installed).  The hot-path modules only need cobaya's logging mixin, so we
pre-seed ``sys.modules`` with a logging stub and register ``eftpipe`` as a bare namespace
rooted at /root/reference/eftpipe.  Nothing is written to the
reference tree (PYTHONDONTWRITEBYTECODE is forced).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("EFTPIPE_REFERENCE", "/root/reference")


def load_reference():
    """Return a namespace with the reference modules used on the hot path."""
    sys.dont_write_bytecode = True
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "eftpipe")):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if "eftpipe" not in sys.modules:
        cob = types.ModuleType("cobaya")
        log = types.ModuleType("cobaya.log")
        mpi = types.ModuleType("cobaya.mpi")

        class HasLogger:
            def set_logger(self, name=None, lowercase=True):
                self._name = name

            def mpi_info(self, *a, **k):
                pass

            mpi_warning = mpi_debug = mpi_info

        class LoggedError(Exception):
            pass

        log.HasLogger = HasLogger
        log.LoggedError = LoggedError
        log.logger_setup = lambda *a, **k: None
        mpi.is_main_process = lambda: True
        mpi.root_only = lambda f: f
        # eftpipe/likelihood.py subclasses cobaya's Likelihood; only its module-level helpers (MultipoleInfo.load, parse_kmask,
        # mask_covariance, hartlap, flatten) are used by the fixture generator, so an empty base class is all the import needs
        lik = types.ModuleType("cobaya.likelihood")

        class Likelihood:
            pass

        lik.Likelihood = Likelihood
        cob.log, cob.mpi, cob.likelihood = log, mpi, lik
        sys.modules.update({"cobaya": cob, "cobaya.log": log, "cobaya.mpi": mpi, "cobaya.likelihood": lik})
        pkg = types.ModuleType("eftpipe")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "eftpipe")]
        sys.modules["eftpipe"] = pkg
    ns = types.SimpleNamespace()
    ns.pybird = importlib.import_module("eftpipe.pybird.pybird")
    ns.fftlog = importlib.import_module("eftpipe.pybird.fftlog")
    ns.resumfactor = importlib.import_module("eftpipe.pybird.resumfactor")
    ns.window = importlib.import_module("eftpipe.window")
    ns.binning = importlib.import_module("eftpipe.binning")
    ns.chained = importlib.import_module("eftpipe.chained")
    ns.parambasis = importlib.import_module("eftpipe.parambasis")
    ns.transformer = importlib.import_module("eftpipe.transformer")
    ns.marginal = importlib.import_module("eftpipe.marginal")
    ns.likelihood = importlib.import_module("eftpipe.likelihood")
    return ns
