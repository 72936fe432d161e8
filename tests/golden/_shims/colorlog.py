"""Import shim: the build container has no `colorlog`; daisy/utils/config.py:7 only
needs ColoredFormatter.  Used ONLY by tests/golden/make_golden.py and the drop-in demo."""
import logging


class ColoredFormatter(logging.Formatter):
    def __init__(self, fmt=None, datefmt=None, log_colors=None, **kw):
        super().__init__(fmt.replace("%(log_color)s", "") if fmt else fmt, datefmt)
