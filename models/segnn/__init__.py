from .l1_tensor_prod import L1TensorProduct  # noqa: F401
