"""Import shim: the build container has no `colorama`; daisy/utils/config.py:9 calls init()."""


def init(*a, **k):
    return None
