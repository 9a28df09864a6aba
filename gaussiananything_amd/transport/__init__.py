from .sampler import Sampler, VelocityTransport, bind_reference_transport, create_transport, integrate  # noqa: F401
