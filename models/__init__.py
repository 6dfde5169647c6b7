"""Drop-in module path of the reference (``models.segnn.l1_tensor_prod``).

Registers the hyphen-named implementation directory ``scalable-e3-gnn_amd/`` as the importable
package ``scalable_e3_gnn_amd``.
"""
import importlib.util
import os
import sys


def _register_package():
    name = "scalable_e3_gnn_amd"
    if name in sys.modules:
        return sys.modules[name]
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scalable-e3-gnn_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "__init__.py"),
                                                  submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[name]
        raise
    return mod


_register_package()
