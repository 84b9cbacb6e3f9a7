"""Package logger.  The reference names its logger "deltamark" (utils/logger.py:23); kept so
that a caller's logging configuration keeps working."""
import logging
import sys

logger = logging.getLogger("deltamark")
if not logger.handlers:
    _h = logging.StreamHandler(sys.stdout)
    _h.setFormatter(logging.Formatter("[%(asctime)s] [%(levelname)s] %(message)s", "%Y-%m-%d %H:%M:%S"))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)
    logger.propagate = False
