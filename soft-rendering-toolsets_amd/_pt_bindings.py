"""ctypes bindings of include/srt_pt.h (path tracer). Filled in with the path-tracer milestone."""


def bind(lib):
    pass


class Scene:  # placeholder until the path-tracer milestone lands
    pass


class Pathtracer:  # placeholder until the path-tracer milestone lands
    pass
