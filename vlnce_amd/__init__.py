"""Import alias: the package directory is named `vln-ce_amd` (not a valid
Python identifier), so `import vlnce_amd` resolves its sub-modules there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vln-ce_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
