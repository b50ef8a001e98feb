"""small host utilities the hot path's callers rely on (mgpi/utils.py)"""

__all__ = ['factory']


def factory(klass):
    """{class name: class} for klass and everything derived from it - how the reference's INI loader finds
    kernels and interpolators by name (mgpi/utils.py:20-26, used at mgpi/io.py:408,537)"""
    found = {klass.__name__: klass}
    pending = list(klass.__subclasses__())
    while pending:
        sub = pending.pop()
        if sub.__name__ not in found:
            found[sub.__name__] = sub
            pending.extend(sub.__subclasses__())
    return found
