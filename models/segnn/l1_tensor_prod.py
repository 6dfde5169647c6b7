"""Same import path as the reference file; the implementation lives in ``scalable-e3-gnn_amd/``."""
from scalable_e3_gnn_amd.l1_tensor_prod import L1TensorProduct  # noqa: F401

__all__ = ["L1TensorProduct"]
