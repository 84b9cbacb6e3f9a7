"""Import alias: `aware.*` resolves to the MI355X implementation in `aware_amd.*`, so code written
against deepmarkpy/aware (`from aware.service import embed_watermark`, `from aware.utils.models import
load`, ...) runs unchanged.  No code lives here."""
import importlib
import sys

import aware_amd as _impl

__version__ = _impl.__version__

for _name in ("interfaces", "utils", "utils.logger", "utils.utils", "utils.audio", "utils.watermark", "utils.models",
              "detection", "embedding", "service", "metrics", "attacks", "pipeline", "parallel"):
    _mod = importlib.import_module("aware_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        setattr(sys.modules[__name__], _name, _mod)
