"""Log-file redirection of the command-line run (reference: run_log.py:12-56): every print() goes,
time-stamped to the millisecond, to `<prefix>.run-log_<timestamp>.txt`."""
from __future__ import annotations

import datetime
import logging
import sys


class _ToLogger(object):
    def __init__(self):
        self._log = logging.getLogger("smcounter_amd.runlog")

    def write(self, buf):
        for line in buf.rstrip().splitlines():
            self._log.debug(line.rstrip())

    def flush(self):
        pass


_saved = None


def init(prefix: str) -> str:
    global _saved
    name = prefix + ".run-log" + datetime.datetime.now().strftime("_%Y.%m.%d_%H.%M.%S") + ".txt"
    log = logging.getLogger("smcounter_amd.runlog")
    log.setLevel(logging.DEBUG)
    h = logging.FileHandler(name, mode="w")
    h.setFormatter(logging.Formatter("%(asctime)s.%(msecs)03d %(message)s", "%Y-%m-%d %H:%M:%S"))
    log.addHandler(h)
    log.propagate = False
    _saved = (sys.stdout, sys.stderr)
    sys.stdout = sys.stderr = _ToLogger()
    return name


def close() -> None:
    global _saved
    if _saved:
        sys.stdout, sys.stderr = _saved
        _saved = None
    log = logging.getLogger("smcounter_amd.runlog")
    for h in list(log.handlers):
        h.close()
        log.removeHandler(h)
