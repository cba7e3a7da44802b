"""Stub of `logbar` so the reference's setup_logger() works without the real package."""
import logging


class _Level:
    def __init__(self, fn):
        self._fn = fn
        self._seen = set()

    def __call__(self, *a, **k):
        return self._fn(*a, **k)

    def once(self, msg, *a, **k):
        if msg in self._seen:
            return
        self._seen.add(msg)
        return self._fn(msg, *a, **k)


class LogBar:
    _shared = None

    def __init__(self):
        lg = logging.getLogger("refshim")
        for name in ("debug", "info", "warning", "error", "critical"):
            setattr(self, name, _Level(getattr(lg, name)))
        self.warn = self.warning

    @classmethod
    def shared(cls, *a, **k):
        if cls._shared is None:
            cls._shared = cls()
        return cls._shared

    def setLevel(self, *a, **k):
        pass

    def pb(self, iterable=None, *a, **k):
        return iterable

    def spinner(self, *a, **k):
        class _S:
            def __enter__(s):
                return s

            def __exit__(s, *e):
                return False

            def close(s):
                pass
        return _S()
