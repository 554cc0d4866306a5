"""Inert stand-in objects: any attribute access, call, item access or arithmetic returns another inert object."""


class Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return Inert()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return Inert()

    def __getitem__(self, k):
        return Inert()

    def __iter__(self):
        return iter(())

    def __bool__(self):
        return False

    def __len__(self):
        return 0

    def __repr__(self):
        return "<inert stand-in>"


class InertMeta(type):
    """classes whose STATIC members are used (`Profiler.start(...)`, `Scene.show(...)`, `Viewer.create(...)`)"""

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return Inert()


def inert_class(name):
    return InertMeta(name, (Inert,), {})
