from .scheduler import MarginScheduler, cosine_decay_with_warmup  # noqa: F401
