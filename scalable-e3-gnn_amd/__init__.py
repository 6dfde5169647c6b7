"""MI355X-native hot path of Scalable-E3-GNN (import name: ``scalable_e3_gnn_amd``).

The directory is called ``scalable-e3-gnn_amd`` (not a valid Python identifier); the repo-root
packages ``models`` / ``__graft_entry__`` / ``tests/conftest.py`` register it under the import name
``scalable_e3_gnn_amd`` via :func:`importlib` (see ``models/__init__.py``).

Public surface:
  * ``L1TensorProduct`` — drop-in for ``models.segnn.l1_tensor_prod.L1TensorProduct`` of the reference.
  * ``Irreps`` / ``Irrep`` / ``Instruction`` — e3nn-shaped bookkeeping (e3nn itself is optional).
Everything computes through ``lib/libe3gnn_hip.so`` (C ABI: ``include/e3gnn.h``); there is no CPU path.
"""
from .irreps import Instruction, Irrep, Irreps, as_blocks  # noqa: F401
from .l1_tensor_prod import L1TensorProduct  # noqa: F401

__all__ = ["L1TensorProduct", "Irreps", "Irrep", "Instruction", "as_blocks"]
