"""Import alias: `import mst_amd` loads the package that lives in `mixing-style-transfer_amd/` (a directory name that
is not a valid Python identifier) through the regular import machinery -- a module spec whose origin and
submodule search path are that directory -- so `mst_amd.model`, relative imports inside the package, `__file__` and
`importlib.reload` all behave as for any package."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mixing-style-transfer_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
