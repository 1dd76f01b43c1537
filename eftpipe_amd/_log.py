"""Logger mixin: cobaya's HasLogger when the engine runs inside eftpipe/cobaya, a minimal
stand-in (the four methods the hot-path classes call) otherwise."""
from __future__ import annotations

import logging

try:  # pragma: no cover - cobaya is not installed in the build image
    from cobaya.log import HasLogger  # type: ignore
except Exception:

    class HasLogger:  # same surface as cobaya.log.HasLogger as used by reference pybird.py / window.py / binning.py
        def set_logger(self, name=None, lowercase=True):
            self.log = logging.getLogger(name or self.__class__.__name__)

        def mpi_info(self, msg, *args, **kwargs):
            self.log.info(msg, *args, **kwargs)

        def mpi_warning(self, msg, *args, **kwargs):
            self.log.warning(msg, *args, **kwargs)

        def mpi_debug(self, msg, *args, **kwargs):
            self.log.debug(msg, *args, **kwargs)
