"""Import alias.  The real package directory is `cuda-bundle-adjustment_amd/` (a dash is not a
legal Python identifier), so `import cuba_amd` forwards to it: submodules resolve there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cuda-bundle-adjustment_amd")
__path__ = [_real]
_init = _os.path.join(_real, "__init__.py")
exec(compile(open(_init).read(), _init, "exec"))
