def setup_logging(*a, **k):
    pass
