"""No-op stand-ins for the two names test_runtime.py:9-10 imports from the third-party ``pytorch_memlab`` package
(it only uses them in commented-out code).  They accept the real package's call shapes and do nothing."""


class LineProfiler:
    def __init__(self, *functions, **kwargs):
        self.functions = functions

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def enable(self):
        pass

    def disable(self):
        pass

    def display(self, *a, **k):
        return ""

    def print_stats(self, *a, **k):
        pass


class MemReporter:
    def __init__(self, model=None, **kwargs):
        self.model = model

    def report(self, *a, **k):
        pass
