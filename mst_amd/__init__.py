"""Import alias: `import mst_amd` loads the package that lives in `mixing-style-transfer_amd/`
(a directory name that is not a valid Python identifier)."""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "mixing-style-transfer_amd")
__path__ = [_pkg_dir]
with open(_os.path.join(_pkg_dir, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg_dir, "__init__.py"), "exec"))
