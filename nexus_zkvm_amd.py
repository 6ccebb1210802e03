"""Import shim: the package directory is named `nexus-zkvm_amd/` (not a valid Python identifier),
so this module turns itself into that package (`import nexus_zkvm_amd` works from the repo root)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "nexus-zkvm_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
