"""Import shim: the package directory is named ``videollm-online_amd`` (not a valid Python
identifier), so this module makes it importable as ``videollm_online_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "videollm-online_amd")]
__package__ = __name__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
