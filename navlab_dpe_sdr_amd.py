"""Import shim: exposes the package directory `navlab-dpe-sdr_amd/` as `navlab_dpe_sdr_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "navlab-dpe-sdr_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _os
