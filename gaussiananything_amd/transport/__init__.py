from .transport import ModelType, PathType, Sampler, SNRType, Transport, WeightType, create_transport  # noqa: F401
