"""stdout INFO logging under the reference's logger name (reference: logger.py:6-10), so the
train/validate/decode log lines look the same to whoever greps them."""
import logging
import sys

from . import info

_configured = False


def get_logger():
    global _configured
    if not _configured:
        logging.basicConfig(stream=sys.stdout, level=logging.INFO)
        _configured = True
    return logging.getLogger(info.app_name)
