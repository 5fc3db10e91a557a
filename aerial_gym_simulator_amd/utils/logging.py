"""Logger shim with the reference's interface (aerial_gym/utils/logging.py:34-53)."""
import logging
import time

_T0 = time.time()


class _ElapsedFormatter(logging.Formatter):
    def format(self, record):
        record.elapsed_ms = int((time.time() - _T0) * 1000)
        return super().format(record)


class CustomLogger(logging.Logger):
    def __init__(self, logger_name):
        super().__init__(logger_name)
        self.setLevel(logging.WARNING)
        handler = logging.StreamHandler()
        handler.setFormatter(_ElapsedFormatter("[%(elapsed_ms)d ms][%(name)s] - %(levelname)s : %(message)s"))
        self.addHandler(handler)

    def setLoggerLevel(self, level):
        self.setLevel(level)

    def print_example_message(self):
        for lvl in ("debug", "info", "warning", "error", "critical"):
            getattr(self, lvl)(f"A {lvl} message")
