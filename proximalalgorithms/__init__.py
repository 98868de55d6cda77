"""Import shim: makes ``import proximalalgorithms.jl_amd`` resolve to the package that lives in the
directory literally named ``proximalalgorithms.jl_amd/`` at the repository root (a dotted directory
name cannot be imported by the default path finder)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "proximalalgorithms.jl_amd")
_name = __name__ + ".jl_amd"
if _name not in _sys.modules:
    _spec = _ilu.spec_from_file_location(_name, _os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
    _mod = _ilu.module_from_spec(_spec)
    _sys.modules[_name] = _mod
    try:
        _spec.loader.exec_module(_mod)
    except BaseException:
        _sys.modules.pop(_name, None)
        raise
jl_amd = _sys.modules[_name]
