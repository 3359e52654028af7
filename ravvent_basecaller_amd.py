"""Import alias: the package directory is ``ravvent-basecaller_amd/`` (the name the build
contract fixes), which Python cannot spell in an ``import`` statement.  This module makes
``import ravvent_basecaller_amd`` resolve to that directory as a regular package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ravvent-basecaller_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
