"""Importable alias of the `latentdiffeq.jl_amd/` package directory (a dot is not legal in a Python
module name). `import latentdiffeq_amd` == the package that lives in ../latentdiffeq.jl_amd/."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "latentdiffeq.jl_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
