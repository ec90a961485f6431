"""Import alias: the package directory is `tiny-ram-halo2_amd/` (not a legal Python
identifier), so `import tiny_ram_halo2_amd` resolves to it through this shim."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tiny-ram-halo2_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
