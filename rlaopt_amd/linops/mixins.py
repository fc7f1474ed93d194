"""``ScaleMixin`` (rlaopt/linops/mixins.py:32-72): multiplies results - or the results of a
wrapped callable - by a constant.  A factor of exactly 1.0 is skipped, nested wrappers
merge their factors."""

__all__ = ["ScaleMixin"]


class _ScaledFunction:
    def __init__(self, fn, scale):
        self.fn, self.scale = fn, scale
        self.__name__ = f"scaled_{getattr(fn, '__name__', 'fn')}"
        self.__doc__ = getattr(fn, "__doc__", None)

    def __call__(self, *args, **kwargs):
        return self.scale * self.fn(*args, **kwargs)


class ScaleMixin:
    def _initialize_scaling(self, scale):
        self._scaling = 1.0 if scale is None else float(scale)

    def _apply_scaling(self, obj):
        factor = getattr(self, "_scaling", 1.0)
        if factor == 1.0:
            return obj
        if isinstance(obj, _ScaledFunction):
            return _ScaledFunction(obj.fn, factor * obj.scale)
        if callable(obj):
            return _ScaledFunction(obj, factor)
        return factor * obj
