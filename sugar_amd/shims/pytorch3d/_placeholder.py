def out_of_scope(name):
    """A class that imports fine and raises when used: mesh rasterization / mesh containers are not rebuilt here."""
    def __init__(self, *a, **k):
        raise NotImplementedError(f"pytorch3d.{name} is outside the scope of the sugar_amd stand-in package "
                                  "(mesh rendering / extraction); install pytorch3d to use it")
    return type(name.rsplit(".", 1)[-1], (), {"__init__": __init__, "__doc__": f"placeholder for pytorch3d.{name}"})


def out_of_scope_fn(name):
    def fn(*a, **k):
        raise NotImplementedError(f"pytorch3d.{name} is outside the scope of the sugar_amd stand-in package; install pytorch3d")
    fn.__name__ = name.rsplit(".", 1)[-1]
    return fn
