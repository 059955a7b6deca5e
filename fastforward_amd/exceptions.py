"""Exception types of the host-side surface (reference: src/fastforward/exceptions.py:5)."""


class QuantizationError(RuntimeError):
    """Raised for quantization-specific failures (strict mode violations, empty dynamic input...)."""


class BackendError(RuntimeError):
    """The HIP backend library is missing or unusable. There is no CPU fallback."""
